"""Host-side mirror of the reference's file classes on the scoring path.

    SolutionFile  -- /root/reference/src/files/SolutionFile.{h,cpp}: the vocabulary
                     (`.tokens.json`), its id assignment, GetId / GetToken / Tokenize
    DataFile      -- /root/reference/src/files/DataFile.{h,cpp}: one corpus shard
    MetadataFile  -- /root/reference/src/files/MetadataFile.{h,cpp}: the corpus index
                     (`.metadata.json`), rebuilt by a directory scan when missing or invalid
    ShardFile     -- the same shard read by the C++ one-pass reader (include/tokfiles.h,
                     csrc/shard_reader.cpp): DataFile's verdict and strings, straight into the
                     packed form, without a DOM
    score_shard   -- /root/reference/src/main.cpp:31-46: the scoring loop over a shard

Same names, argument meaning and error behaviour as the reference: a file that cannot be
read or does not validate leaves the object with IsValid() == False, nothing is thrown
(JsonFile.h:17, JsonFile.cpp:14-29).  All segmentation runs on the GPU through the C-ABI
(libtokscore.so); there is no CPU path in here.
"""
import json
import re
import os

import numpy as np

from . import Scorer

BUILD_VERSION = "0.4.0"  # /root/reference/src/config.h:3
START_TOKEN = "<START>"  # SolutionFile.cpp:10
END_TOKEN = "<END>"      # SolutionFile.cpp:11
UNKNOWN = (1 << 64) - 1  # (size_t)-1, SolutionFile.cpp:76,93

_SHORT_ESCAPES = {0x08: "\\b", 0x0C: "\\f", 0x0A: "\\n", 0x0D: "\\r", 0x09: "\\t", 0x22: '\\"', 0x5C: "\\\\"}


def _json_string(s: str) -> str:
    """A JSON string exactly as rapidjson's Writer prints it (writer.h WriteString): only
    '"', '\\' and control characters are escaped, the rest is raw UTF-8."""
    out = ['"']
    for ch in s:
        o = ord(ch)
        if o in _SHORT_ESCAPES:
            out.append(_SHORT_ESCAPES[o])
        elif o < 0x20:
            out.append("\\u%04X" % o)
        else:
            out.append(ch)
    out.append('"')
    return "".join(out)


def _first_wins(pairs):
    """rapidjson looks a member up with FindMember: the FIRST one of a repeated name."""
    d = {}
    for k, v in pairs:
        d.setdefault(k, v)
    return d


def _no_constant(name):  # NaN / Infinity are not JSON; rapidjson's default flags reject them
    raise ValueError(name)


def _finite_float(text):  # kParseErrorNumberTooBig: a number that does not fit a double
    v = float(text)
    if v in (float("inf"), float("-inf")):
        raise ValueError(text)
    return v


_ESCAPE = re.compile(r"\\(?:u([0-9a-fA-F]{4})|.)", re.S)


def _surrogate_escapes_paired(text: str) -> bool:
    """rapidjson rejects a \\uD800-\\uDBFF escape that is not followed by a \\uDC00-\\uDFFF one, and a
    low one on its own (kParseErrorStringUnicodeSurrogateInvalid); Python's json lets both through."""
    pos = 0
    while True:
        m = _ESCAPE.search(text, pos)
        if not m:
            return True
        pos = m.end()
        if m.group(1) is None:
            continue
        cp = int(m.group(1), 16)
        if 0xDC00 <= cp <= 0xDFFF:
            return False
        if 0xD800 <= cp <= 0xDBFF:
            low = _ESCAPE.match(text, pos)
            if not low or low.group(1) is None or not 0xDC00 <= int(low.group(1), 16) <= 0xDFFF:
                return False
            pos = low.end()


def _load_json(path):
    """The document as rapidjson's default flags read it (JsonFile.cpp:21-27): UTF-8 is NOT validated,
    bytes that are not UTF-8 pass through strings as they are (here as surrogate escapes, which _utf8
    turns back into the same bytes).  None for whatever rapidjson rejects or cannot be read."""
    try:
        with open(path, "rb") as f:
            text = f.read().decode("utf-8", errors="surrogateescape")
        if not _surrogate_escapes_paired(text):
            return None
        return json.loads(text, object_pairs_hook=_first_wins, parse_constant=_no_constant, parse_float=_finite_float)
    except (OSError, ValueError, RecursionError):  # (rapidjson's own parser is iterative; a document nested too deep for this one is given up on)
        return None


def _utf8(s: str):
    try:
        return s.encode("utf-8", errors="surrogateescape")
    except UnicodeEncodeError:  # a lone surrogate that is not an escaped byte
        return None


class SolutionFile:
    """SolutionFile(path) reads a `.tokens.json`; SolutionFile(tokens, path) assigns ids to
    `tokens` and writes the file (SolutionFile.cpp:43-72).  Ids: 0 = <START>, 1 = <END>,
    token i = i + 2; a repeated string keeps its first id (:56,69)."""

    def __init__(self, tokens_or_path, path=None, device: int = 0):
        self._device = device
        self._scorer = None
        self._ids = {}
        self.tokens = []  # bytes, in file order
        self.max_len_ = 0
        self.valid_ = False
        if path is None:
            self.path_ = tokens_or_path
            doc = _load_json(self.path_)
            if not self._validate(doc):
                return
            toks = [_utf8(t) for t in doc["tokens"]]
            if any(t is None for t in toks):
                return
            self.tokens = toks
            self.valid_ = True
        else:
            self.path_ = path
            self.tokens = [t if isinstance(t, bytes) else t.encode("utf-8") for t in tokens_or_path]
            self._build_doc()
            # (the reference leaves valid_ false on this constructor, JsonFile.cpp:16-17, and uses
            # the object all the same, main.cpp:26-46)
        for i, t in enumerate(self.tokens):
            self.max_len_ = max(self.max_len_, len(t))
            self._ids.setdefault(t, i + 2)

    @staticmethod
    def _validate(doc) -> bool:  # SolutionFile.cpp:14-26
        if not isinstance(doc, dict):
            return False
        if not isinstance(doc.get("version"), str) or doc["version"] != BUILD_VERSION:
            return False
        if not isinstance(doc.get("tokens"), list):
            return False
        return all(isinstance(t, str) for t in doc["tokens"])

    def _build_doc(self):  # SolutionFile.cpp:27-41 + JsonFile::Save (JsonFile.cpp:35-46)
        # (rapidjson's Writer copies bytes >= 0x80 as they are, valid UTF-8 or not)
        body = ", ".join(_json_string(t.decode("utf-8", "surrogateescape")) for t in self.tokens)
        text = '{\n    "version": %s,\n    "tokens": [%s]\n}' % (_json_string(BUILD_VERSION), body)
        with open(self.path_, "wb") as f:
            f.write(text.encode("utf-8", "surrogateescape"))

    def IsValid(self) -> bool:
        return self.valid_

    def GetId(self, token) -> int:  # SolutionFile.cpp:74-77
        t = token if isinstance(token, bytes) else token.encode("utf-8")
        return self._ids.get(t, UNKNOWN)

    def GetToken(self, id_: int) -> str:  # SolutionFile.cpp:79-81
        if id_ == UNKNOWN:
            return "<UNKNOWN>"
        if id_ == 0:
            return START_TOKEN
        if id_ == 1:
            return END_TOKEN
        return self.tokens[id_ - 2].decode("utf-8", errors="replace")

    def _token_c_str(self, id_: int) -> bytes:
        # GetToken hands out a C string (SolutionFile.cpp:79-81): whoever appends it stops at a NUL
        if id_ == UNKNOWN:
            return b"<UNKNOWN>"
        if id_ == 0:
            return START_TOKEN.encode()
        if id_ == 1:
            return END_TOKEN.encode()
        return self.tokens[id_ - 2].split(b"\x00", 1)[0]

    def Detokenize(self, ids) -> bytes:  # SolutionFile.cpp:102-108
        """The tokens' strings back to back, <START> / <END> / <UNKNOWN> spelled out -- bytes, as the
        reference's std::string is (a token may end inside a UTF-8 sequence)."""
        return b"".join(self._token_c_str(i) for i in ids)

    def Prettify(self, ids) -> bytes:  # SolutionFile.cpp:110-118
        """The same with '|' between the tokens (the reference calls pop_back() on an empty string
        for an empty id list, which is undefined: here that is the empty string)."""
        return b"|".join(self._token_c_str(i) for i in ids)

    def scorer(self) -> Scorer:
        """The GPU context that holds this vocabulary (created on first use)."""
        if self._scorer is None:
            s = Scorer(self._device)
            s.set_vocab_list(self.tokens)
            self._scorer = s
        return self._scorer

    def Tokenize(self, text):  # SolutionFile.cpp:83-100, on the GPU
        data = text if isinstance(text, bytes) else text.encode("utf-8")
        return [UNKNOWN if i < 0 else i for i in self.scorer().tokenize(data)]

    def close(self):
        if self._scorer is not None:
            self._scorer.close()
            self._scorer = None


class DataFile:
    """One corpus shard: a JSON array of {"id", "title", "text"} objects (DataFile.cpp:3-22)."""

    class Entry:
        __slots__ = ("id", "title", "text")

        def __init__(self, id_, title, text):
            self.id, self.title, self.text = id_, title, text

    def __init__(self, path):
        self.path_ = path
        self.valid_ = False
        self._entries = []
        doc = _load_json(path)
        if not self._validate(doc):
            return
        entries = []
        for e in doc:
            # GetString() hands a C string to std::string: cut at the first NUL (DataFile.cpp:27-29)
            fields = [_utf8(e[k].split("\x00", 1)[0]) for k in ("id", "title", "text")]
            if any(f is None for f in fields):
                return
            entries.append(DataFile.Entry(*fields))
        self._entries = entries
        self.valid_ = True

    @staticmethod
    def _validate(doc) -> bool:  # DataFile.cpp:3-17
        if not isinstance(doc, list) or not doc:
            return False
        for e in doc:
            if not isinstance(e, dict):
                return False
            if not all(isinstance(e.get(k), str) for k in ("id", "text", "title")):
                return False
        return True

    def IsValid(self) -> bool:
        return self.valid_

    def GetEntries(self):  # DataFile.cpp:24-31 (owning copies)
        return list(self._entries)

    def packed(self):
        """The shard as the C-ABI takes it: document bytes back to back + D + 1 offsets."""
        texts = [e.text for e in self._entries]
        off = np.zeros(len(texts) + 1, dtype=np.uint64)
        if texts:
            off[1:] = np.cumsum([len(t) for t in texts], dtype=np.uint64)
        corpus = np.frombuffer(b"".join(texts), dtype=np.uint8) if off[-1] else np.zeros(0, dtype=np.uint8)
        return corpus, off


class MetadataFile:
    """`.metadata.json` = {"version", "files": [{"path"}]} next to the shards (MetadataFile.cpp:17-75).
    MetadataFile(path, rebuild=False): read and validate; when the file is missing, invalid or
    rebuild is set, every `*.json` below the file's directory that is a valid DataFile is listed
    (paths relative to that directory) and the file is written.  The reference lists them in
    thread-completion order (MetadataFile.cpp:46-62, not deterministic); here they are sorted."""

    class Entry:
        __slots__ = ("path",)

        def __init__(self, path):
            self.path = path

    def __init__(self, path, rebuild: bool = False):
        self.path_ = path
        self.modified_ = False
        doc = None if rebuild else _load_json(path)  # JsonFile(path, rebuild) (JsonFile.cpp:14-29)
        self.valid_ = self._validate(doc)
        if self.valid_:
            self._files = [e["path"] for e in doc["files"]]
        else:
            self._build_doc()

    @staticmethod
    def _validate(doc) -> bool:  # MetadataFile.cpp:17-31
        if not isinstance(doc, dict):
            return False
        if not isinstance(doc.get("version"), str) or doc["version"] != BUILD_VERSION:
            return False
        if not isinstance(doc.get("files"), list):
            return False
        return all(isinstance(e, dict) and isinstance(e.get("path"), str) for e in doc["files"])

    def _build_doc(self):  # MetadataFile.cpp:33-69 + JsonFile::Save (JsonFile.cpp:35-46)
        self.modified_ = True
        root = self.GetRootPath()
        found = []
        for base, _, names in os.walk(root):
            for name in names:
                full = os.path.join(base, name)
                if os.path.splitext(name)[1] != ".json" or not os.path.isfile(full):
                    continue
                if DataFile(full).IsValid():
                    found.append(os.path.relpath(full, root))
        self._files = sorted(found)
        # PrettyWriter with kFormatSingleLineArray: the array stays on one line, the objects in it do not
        objs = ", ".join('{\n            "path": %s\n        }' % _json_string(p) for p in self._files)
        body = "[%s]" % objs
        text = '{\n    "version": %s,\n    "files": %s\n}' % (_json_string(BUILD_VERSION), body)
        with open(self.path_, "wb") as f:
            f.write(text.encode("utf-8"))

    def IsValid(self) -> bool:
        # (as in the reference, a rebuilt file leaves valid_ false: MetadataFile.cpp:73-74)
        return self.valid_

    def GetRootPath(self) -> str:  # MetadataFile.cpp:77-79
        return os.path.realpath(os.path.dirname(os.path.abspath(self.path_)))

    def GetFiles(self, file_cnt: int = -1):  # MetadataFile.cpp:80-87
        files = []
        for p in self._files:
            files.append(MetadataFile.Entry(p))
            file_cnt -= 1
            if file_cnt == 0:
                break
        return files


class ShardFile:
    """DataFile(path) through libtokanneal.so's one-pass reader (tokfiles_shard_read): IsValid(),
    GetEntries() and packed() as DataFile's, for shards too large to want a JSON DOM of."""

    def __init__(self, path):
        import ctypes as C
        from . import annealing
        lib = annealing.load_library()
        self.path_ = path
        h = C.c_void_p()
        valid = C.c_int(0)
        st = lib.tokfiles_shard_read(C.byref(h), os.fsencode(path), C.byref(valid))
        if st != 0:
            raise MemoryError("tokfiles_shard_read: status %d" % st)
        self.valid_ = bool(valid.value)
        self._fields = []
        try:
            n = int(lib.tokfiles_shard_docs(h))
            for field in range(3):  # text, id, title
                b = np.zeros(max(1, int(lib.tokfiles_shard_bytes(h, field))), dtype=np.uint8)
                off = np.zeros(n + 1, dtype=np.uint64)
                lib.tokfiles_shard_copy(h, field, b.ctypes.data_as(C.POINTER(C.c_uint8)), off.ctypes.data_as(C.POINTER(C.c_uint64)))
                self._fields.append((b[:int(off[-1])], off))
        finally:
            lib.tokfiles_shard_free(h)

    def IsValid(self) -> bool:
        return self.valid_

    def packed(self):
        return self._fields[0]

    def GetEntries(self):
        cut = []
        for b, off in self._fields:
            raw = b.tobytes()
            cut.append([raw[int(off[i]):int(off[i + 1])] for i in range(len(off) - 1)])
        return [DataFile.Entry(i, t, x) for x, i, t in zip(*cut)]


def score_shard(solution: SolutionFile, shard: DataFile):
    """main.cpp:31-46 over one shard: (counts[V+3], total_tokens = comp_size, total_bytes =
    init_size); counts[id] for ids 0..V+1, counts[V+2] = UNKNOWN."""
    s = solution.scorer()
    s.set_corpus(*shard.packed())
    return s.score()
