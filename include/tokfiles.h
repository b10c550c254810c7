/* tokfiles.h -- C ABI for reading a corpus shard straight into the packed form the scorer takes
 * (a streaming shard reader).  Exported by libtokanneal.so (host code,
 * no GPU involved); implementation in tokenizer_amd/csrc/shard_reader.cpp.
 *
 * Replaces, on the corpus-loading side of the scoring path,
 *   DataFile::DataFile(path) + IsValid()   src/files/DataFile.cpp:3-22, src/files/JsonFile.cpp:14-29
 *   DataFile::GetEntries()                 src/files/DataFile.cpp:24-31
 * -- one pass over the file instead of a rapidjson DOM plus three string copies per entry.  Same
 * verdict and same strings as the reference: strict JSON (rapidjson default parse flags), root =
 * non-empty array of objects whose first "id", "text" and "title" members are strings, every
 * string cut at its first NUL (GetString() is taken as a C string, DataFile.cpp:27-29). */
#ifndef TOKFILES_H
#define TOKFILES_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TOKFILES_OK 0
#define TOKFILES_E_ARG 1
#define TOKFILES_E_MEMORY 2

#define TOKFILES_TEXT 0  /* DataFile::Entry::text  -- what is scored */
#define TOKFILES_ID 1    /* DataFile::Entry::id */
#define TOKFILES_TITLE 2 /* DataFile::Entry::title */

typedef struct tokfiles_shard tokfiles_shard;

/* DataFile(path): reads and validates.  *valid = IsValid() (0 for a file that cannot be opened,
 * does not parse or is not a shard: never an error status, JsonFile.h:17).  An invalid shard
 * has no documents. */
int tokfiles_shard_read(tokfiles_shard **out, const char *path, int *valid);

/* number of entries; bytes of one field over all entries */
uint64_t tokfiles_shard_docs(const tokfiles_shard *s);
uint64_t tokfiles_shard_bytes(const tokfiles_shard *s, int field);

/* GetEntries(), one field, packed: entry d = bytes[off[d] .. off[d+1]); bytes has room for
 * tokfiles_shard_bytes(s, field), off for docs + 1 entries.  TOKFILES_TEXT gives exactly the
 * arguments of tokscore_set_corpus. */
int tokfiles_shard_copy(const tokfiles_shard *s, int field, uint8_t *bytes, uint64_t *off);

void tokfiles_shard_free(tokfiles_shard *s);

#ifdef __cplusplus
}
#endif
#endif
