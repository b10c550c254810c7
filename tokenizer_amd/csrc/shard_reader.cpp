// shard_reader.cpp -- a corpus shard (JSON array of {"id","title","text"}) read in one pass straight
// into the packed form the GPU scorer takes ("a streaming shard reader to
// feed GPUs faster than rapidjson DOM").  Replaces, for this path,
//   JsonFile::JsonFile(path, false)   src/files/JsonFile.cpp:14-29   (rapidjson DOM of the whole file)
//   DataFile::Validate                src/files/DataFile.cpp:3-17
//   DataFile::GetEntries              src/files/DataFile.cpp:24-31   (three std::string copies per entry)
// with the same verdict (IsValid) and the same strings: strict JSON as rapidjson's default flags
// parse it (no comments, no trailing commas, one root value, control characters and lone
// surrogates in strings are errors, UTF-8 is not validated), the FIRST member of each name
// counts (GenericValue::FindMember), and every string is cut at its first NUL because the
// reference hands GetString() to std::string as a C string (DataFile.cpp:27-29).
// Host code, no GPU involved; exported from libtokanneal.so (include/tokfiles.h).
#include "../../include/tokfiles.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

struct Field {
	std::vector<uint8_t> bytes;
	std::vector<uint64_t> off{0};
};

constexpr int kMaxDepth = 4096; // (rapidjson recurses without a bound)

class Parser {
public:
	Parser(const uint8_t *begin, const uint8_t *end) : p_(begin), end_(end) {}

	// the whole document; false = parse error or not a shard (DataFile::Validate)
	bool ParseShard(Field f[3]) {
		SkipWs();
		if (!Take('[')) return false; // (any other root is "not an array" or a parse error: invalid both ways)
		SkipWs();
		if (Peek() == ']') return false; // doc_.Empty() (DataFile.cpp:6); the rest may even be malformed
		uint64_t entries = 0;
		for (;;) {
			SkipWs();
			if (!ParseEntry(f)) return false;
			entries++;
			SkipWs();
			if (Take(',')) continue;
			if (Take(']')) break;
			return false;
		}
		SkipWs();
		return AtEnd() && entries > 0; // kParseErrorDocumentRootNotSingular otherwise
	}

private:
	const uint8_t *p_, *end_;
	std::string key_;

	bool AtEnd() const { return p_ >= end_ || *p_ == 0; } // FileReadStream: a zero byte reads as the end
	int Peek() const { return AtEnd() ? 0 : *p_; }
	bool Take(uint8_t c) {
		if (p_ < end_ && *p_ == c) {
			p_++;
			return true;
		}
		return false;
	}
	void SkipWs() {
		while (p_ < end_ && (*p_ == ' ' || *p_ == '\n' || *p_ == '\r' || *p_ == '\t')) p_++;
	}

	static int Hex(uint8_t c) {
		if (c >= '0' && c <= '9') return c - '0';
		if (c >= 'a' && c <= 'f') return c - 'a' + 10;
		if (c >= 'A' && c <= 'F') return c - 'A' + 10;
		return -1;
	}
	bool Hex4(unsigned *cp) {
		if (end_ - p_ < 4) return false;
		unsigned v = 0;
		for (int k = 0; k < 4; k++) {
			const int h = Hex(p_[k]);
			if (h < 0) return false;
			v = v << 4 | (unsigned)h;
		}
		p_ += 4;
		*cp = v;
		return true;
	}
	template <typename Out> static void Encode(unsigned cp, Out *out) { // UTF8<>::Encode
		if (cp < 0x80) {
			out->push_back((char)cp);
		} else if (cp < 0x800) {
			out->push_back((char)(0xC0 | cp >> 6));
			out->push_back((char)(0x80 | (cp & 0x3F)));
		} else if (cp < 0x10000) {
			out->push_back((char)(0xE0 | cp >> 12));
			out->push_back((char)(0x80 | (cp >> 6 & 0x3F)));
			out->push_back((char)(0x80 | (cp & 0x3F)));
		} else {
			out->push_back((char)(0xF0 | cp >> 18));
			out->push_back((char)(0x80 | (cp >> 12 & 0x3F)));
			out->push_back((char)(0x80 | (cp >> 6 & 0x3F)));
			out->push_back((char)(0x80 | (cp & 0x3F)));
		}
	}

	// first byte in [p_, end_) that is '"', '\\' or below 0x20, eight bytes at a time
	const uint8_t *ScanPlain() const {
		const uint8_t *q = p_;
		while (end_ - q >= 8) {
			uint64_t w;
			memcpy(&w, q, 8);
			const uint64_t k01 = 0x0101010101010101ull, k80 = 0x8080808080808080ull;
			const uint64_t quote = w ^ (k01 * '"'), slash = w ^ (k01 * '\\');
			const uint64_t hit = ((quote - k01) & ~quote) | ((slash - k01) & ~slash) | ((w - k01 * 0x20) & ~w);
			if (hit & k80) break;
			q += 8;
		}
		while (q < end_ && *q != '"' && *q != '\\' && *q >= 0x20) q++;
		return q;
	}

	// a string, opening quote already taken; decoded into *out when out is not null
	template <typename Out> bool ParseString(Out *out) {
		for (;;) {
			const uint8_t *q = ScanPlain();
			if (out && q > p_) out->insert(out->end(), p_, q);
			p_ = q;
			if (p_ >= end_) return false;
			const uint8_t c = *p_++;
			if (c == '"') return true;
			if (c != '\\') return false; // control character (or the zero byte) inside a string
			if (p_ >= end_) return false;
			const uint8_t e = *p_++;
			unsigned cp;
			switch (e) {
			case '"': cp = '"'; break;
			case '\\': cp = '\\'; break;
			case '/': cp = '/'; break;
			case 'b': cp = '\b'; break;
			case 'f': cp = '\f'; break;
			case 'n': cp = '\n'; break;
			case 'r': cp = '\r'; break;
			case 't': cp = '\t'; break;
			case 'u': {
				if (!Hex4(&cp)) return false;
				if (cp >= 0xD800 && cp <= 0xDFFF) { // reader.h:1026-1043
					if (cp > 0xDBFF) return false;
					unsigned lo;
					if (!Take('\\') || !Take('u') || !Hex4(&lo)) return false;
					if (lo < 0xDC00 || lo > 0xDFFF) return false;
					cp = (((cp - 0xD800) << 10) | (lo - 0xDC00)) + 0x10000;
				}
				break;
			}
			default: return false;
			}
			if (out) Encode(cp, out);
		}
	}

	bool ParseNumber() {
		const uint8_t *s = p_;
		Take('-');
		if (Take('0')) {
		} else if (p_ < end_ && *p_ >= '1' && *p_ <= '9') {
			while (p_ < end_ && *p_ >= '0' && *p_ <= '9') p_++;
		} else {
			return false;
		}
		bool real = false;
		if (Take('.')) {
			real = true;
			if (!(p_ < end_ && *p_ >= '0' && *p_ <= '9')) return false;
			while (p_ < end_ && *p_ >= '0' && *p_ <= '9') p_++;
		}
		if (p_ < end_ && (*p_ == 'e' || *p_ == 'E')) {
			real = true;
			p_++;
			if (p_ < end_ && (*p_ == '+' || *p_ == '-')) p_++;
			if (!(p_ < end_ && *p_ >= '0' && *p_ <= '9')) return false;
			while (p_ < end_ && *p_ >= '0' && *p_ <= '9') p_++;
		}
		if (real || p_ - s > 19) { // kParseErrorNumberTooBig: the value does not fit a double
			const std::string t((const char *)s, (size_t)(p_ - s));
			if (std::isinf(strtod(t.c_str(), nullptr))) return false;
		}
		return true;
	}

	bool Literal(const char *w) {
		const size_t n = strlen(w);
		if ((size_t)(end_ - p_) < n || memcmp(p_, w, n) != 0) return false;
		p_ += n;
		return true;
	}

	// any value, checked and thrown away
	bool SkipValue(int depth) {
		if (depth > kMaxDepth) return false;
		switch (Peek()) {
		case '"': p_++; return ParseString<std::string>(nullptr);
		case '{': {
			p_++;
			SkipWs();
			if (Take('}')) return true;
			for (;;) {
				SkipWs();
				if (!Take('"') || !ParseString<std::string>(nullptr)) return false;
				SkipWs();
				if (!Take(':')) return false;
				SkipWs();
				if (!SkipValue(depth + 1)) return false;
				SkipWs();
				if (Take(',')) continue;
				return Take('}');
			}
		}
		case '[': {
			p_++;
			SkipWs();
			if (Take(']')) return true;
			for (;;) {
				SkipWs();
				if (!SkipValue(depth + 1)) return false;
				SkipWs();
				if (Take(',')) continue;
				return Take(']');
			}
		}
		case 't': return Literal("true");
		case 'f': return Literal("false");
		case 'n': return Literal("null");
		default: return ParseNumber();
		}
	}

	// one element of the root array: an object whose first "id", "text" and "title" members are strings
	bool ParseEntry(Field f[3]) {
		if (!Take('{')) return false; // !entry.IsObject() (or malformed)
		bool seen[3] = {false, false, false};
		SkipWs();
		if (!Take('}')) {
			for (;;) {
				SkipWs();
				if (!Take('"')) return false;
				key_.clear();
				if (!ParseString(&key_)) return false;
				SkipWs();
				if (!Take(':')) return false;
				SkipWs();
				const int which = key_ == "text" ? 0 : key_ == "id" ? 1 : key_ == "title" ? 2 : -1;
				if (which >= 0 && !seen[which]) {
					seen[which] = true;
					if (!Take('"')) return false; // present but not a string (DataFile.cpp:10,12,14)
					Field &dst = f[which];
					const size_t start = dst.bytes.size();
					if (!ParseString(&dst.bytes)) return false;
					// std::string(const char*) stops at the first NUL (DataFile.cpp:27-29)
					const void *nul = memchr(dst.bytes.data() + start, 0, dst.bytes.size() - start);
					if (nul) dst.bytes.resize((const uint8_t *)nul - dst.bytes.data());
				} else if (!SkipValue(1)) {
					return false;
				}
				SkipWs();
				if (Take(',')) continue;
				if (Take('}')) break;
				return false;
			}
		}
		if (!(seen[0] && seen[1] && seen[2])) return false; // HasMember (DataFile.cpp:9,11,13)
		for (int k = 0; k < 3; k++) f[k].off.push_back(f[k].bytes.size());
		return true;
	}
};

} // namespace

struct tokfiles_shard {
	bool valid = false;
	Field f[3]; // text, id, title
};

extern "C" {

int tokfiles_shard_read(tokfiles_shard **out, const char *path, int *valid) {
	if (!out || !path) return TOKFILES_E_ARG;
	try {
		auto s = std::make_unique<tokfiles_shard>();
		// (a file that cannot be opened is an invalid DataFile, not an error: JsonFile.cpp:19-20)
		const int fd = open(path, O_RDONLY);
		if (fd >= 0) {
			struct stat st;
			if (fstat(fd, &st) == 0 && st.st_size > 0) {
				const size_t n = (size_t)st.st_size;
				// the file is parsed where the page cache holds it; the text field gets its only
				// allocation up front (decoded text is never longer than the JSON that spells it)
				void *map = mmap(nullptr, n, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
				if (map != MAP_FAILED) {
					(void)madvise(map, n, MADV_SEQUENTIAL);
					s->f[0].bytes.reserve(n);
					Parser ps((const uint8_t *)map, (const uint8_t *)map + n);
					s->valid = ps.ParseShard(s->f);
					munmap(map, n);
				}
			}
			close(fd);
		}
		if (!s->valid)
			for (Field &f : s->f) f = Field();
		if (valid) *valid = s->valid ? 1 : 0;
		*out = s.release();
		return TOKFILES_OK;
	} catch (...) {
		return TOKFILES_E_MEMORY;
	}
}

uint64_t tokfiles_shard_docs(const tokfiles_shard *s) { return s ? s->f[0].off.size() - 1 : 0; }

uint64_t tokfiles_shard_bytes(const tokfiles_shard *s, int field) {
	return s && field >= 0 && field < 3 ? s->f[field].bytes.size() : 0;
}

int tokfiles_shard_copy(const tokfiles_shard *s, int field, uint8_t *bytes, uint64_t *off) {
	if (!s || field < 0 || field > 2 || !off) return TOKFILES_E_ARG;
	const Field &f = s->f[field];
	if (!f.bytes.empty()) {
		if (!bytes) return TOKFILES_E_ARG;
		memcpy(bytes, f.bytes.data(), f.bytes.size());
	}
	memcpy(off, f.off.data(), f.off.size() * sizeof(uint64_t));
	return TOKFILES_OK;
}

void tokfiles_shard_free(tokfiles_shard *s) { delete s; }

} // extern "C"
