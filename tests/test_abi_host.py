"""CPU-side checks: the C-ABI library loads, exports every symbol include/tokscore.h
declares, fails loudly without a GPU, and the host-side packed trie (the double array the
kernels walk) agrees with the oracle's exact-string map on every golden vocabulary."""
import os
import re

import numpy as np
import pytest

import oracle_ffi as O
import tokenizer_amd as ta
from tokenizer_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = ta.load_library()
    header = open(os.path.join(ROOT, "include", "tokscore.h")).read()
    declared = set(re.findall(r"\b(tokscore_[a-z_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), "libtokscore.so does not export " + name
    assert declared == set(ta.EXPORTED_SYMBOLS)
    assert lib.tokscore_abi_version() == 100


def test_no_gpu_means_loud_failure_not_fallback():
    if ta.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(ta.TokscoreError) as e:
        ta.Scorer(0)
    assert e.value.status == ta.E_NO_DEVICE
    s = ta.Scorer(host_only=True)
    s.set_vocab_list([b"a", b"ab"])
    with pytest.raises(ta.TokscoreError) as e:
        s.score()
    assert e.value.status == ta.E_NO_DEVICE
    with pytest.raises(ta.TokscoreError) as e:
        s.set_corpus(np.zeros(4, dtype=np.uint8), np.array([0, 4], dtype=np.uint64))
    assert e.value.status == ta.E_NO_DEVICE


def test_vocab_without_a_nonempty_token_is_rejected():
    s = ta.Scorer(host_only=True)
    with pytest.raises(ta.TokscoreError) as e:
        s.set_vocab_list([b"", b""])
    assert e.value.status == ta.E_VOCAB
    with pytest.raises(ta.TokscoreError):
        s.set_vocab_list([])


def _longest_by_oracle(voc, text):
    """The reference rule (SolutionFile.cpp:90-96): first hit scanning len = min(max_len, n)..1."""
    low = bytes(c + 32 if 65 <= c <= 90 else c for c in text)
    for ln in range(min(voc.max_len, len(low)), 0, -1):
        tid = voc.get_id(low[:ln])
        if tid != -1:
            return ln, tid
    return 0, 0


@pytest.mark.parametrize("case", O.golden_cases())
def test_packed_trie_matches_exact_string_map(case):
    g = O.load_golden(case)
    tb, to = synth.pack_vocab(g["tokens"])
    voc = O.OracleVocab(tb, to)
    s = ta.Scorer(host_only=True)
    s.set_vocab(tb, to)
    assert s.max_len == voc.max_len
    # every token string, and every proper prefix / extension of a few of them
    rng = np.random.default_rng(1)
    toks = g["tokens"]
    pick = toks if len(toks) < 400 else [toks[i] for i in rng.choice(len(toks), 400, replace=False)]
    for t in pick:
        for probe in (t, t[:-1], t + b"q", t + t):
            if probe:
                assert s.hosttrie_longest(probe) == _longest_by_oracle(voc, probe), (case, probe)
    # windows of the documents themselves (capital letters included)
    text = b"".join(g["docs"])[:20000]
    for p in rng.integers(0, max(1, len(text)), 300).tolist():
        w = text[p:p + voc.max_len + 3]
        if w:
            assert s.hosttrie_longest(w) == _longest_by_oracle(voc, w), (case, p)


def test_toggle_on_host_trie_promotes_duplicates():
    s = ta.Scorer(host_only=True)
    s.set_vocab_list([b"a", b"ab", b"ab", b"abc"])
    assert s.hosttrie_longest(b"abx") == (2, 3)
    s.toggle(1, False)   # first "ab" off -> the duplicate (index 2, id 4) takes over
    assert s.hosttrie_longest(b"abx") == (2, 4)
    s.toggle(2, False)
    assert s.hosttrie_longest(b"abx") == (1, 2)
    assert s.hosttrie_longest(b"abc") == (3, 5)
    s.toggle(1, True)
    assert s.hosttrie_longest(b"abx") == (2, 3)


def test_every_build_command_compiles_device_code_for_gfx950():
    """What broke round 1's one-command build: with `-x c++` anywhere on its line hipcc drops
    --offload-arch and builds the device code for its default (gfx906), so the library holds no
    gfx950 code object and the first launch fails.  The commands the Makefile runs must all say
    gfx950 to the device compiler (the driver's own plan, nothing is compiled here)."""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    csrc = os.path.join(ROOT, "tokenizer_amd", "csrc")
    dry = subprocess.run(["make", "-C", csrc, "-n", "-B", "all", "single"], capture_output=True, text=True, check=True).stdout
    dry = dry.replace("\\\n", " ")
    cmds = [ln.strip() for ln in dry.splitlines() if "hipcc" in ln and ".hip" in ln]
    assert len(cmds) >= 7, dry
    for cmd in cmds:
        assert "-x c++" not in cmd, cmd
        r = subprocess.run(cmd.split() + ["-###"], capture_output=True, text=True, cwd=csrc)
        archs = set(re.findall(r"mcpu=(gfx\w+)", r.stderr + r.stdout))
        assert archs == {"gfx950"}, (cmd, archs)
