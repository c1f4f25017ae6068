// ref_harness.cpp -- thin C ABI over the two reference sources that compile on their own:
//   /root/reference/src/kseq.h       (FASTA reader used at src/SketchInfo.cpp:880-948)
//   /root/reference/src/UnionFind.h  (Kruskal's union-find, src/MST.cpp:59-75)
// Built by oracle/Makefile with -I/root/reference/src into oracle/_ref/ (git-ignored).  The
// reference sources are included from where they lie; nothing is copied into this repo.
// TEST INFRASTRUCTURE ONLY: used to pin the oracle / host FASTA reader in this container and to
// generate tests/golden fixtures.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include "UnionFind.h"
#include "kseq.h"
KSEQ_INIT(gzFile, gzread)

extern "C" {

// Parses `path` with kseq exactly like sketchFiles does and writes, per record:
// name\tcomment\tlength\n followed by the raw sequence and '\n' into out (caller-sized).
// Returns the number of bytes needed (call twice) or -1 on open failure.
long ref_kseq_dump(const char* path, char* out, long cap) {
  gzFile fp = gzopen(path, "r");
  if (!fp) return -1;
  kseq_t* ks = kseq_init(fp);
  long pos = 0;
  auto put = [&](const char* s, long n) {
    if (out && pos + n <= cap) memcpy(out + pos, s, n);
    pos += n;
  };
  while (1) {
    int length = kseq_read(ks);
    if (length < 0) break;
    const char* name = ks->name.s ? ks->name.s : "noName";        // src/SketchInfo.cpp:934-939
    const char* comment = ks->comment.s ? ks->comment.s : "noName";
    char num[32];
    int nn = snprintf(num, sizeof num, "%d", length);
    put(name, (long)strlen(name)); put("\t", 1);
    put(comment, (long)strlen(comment)); put("\t", 1);
    put(num, nn); put("\n", 1);
    put(ks->seq.s, length); put("\n", 1);
  }
  kseq_destroy(ks);
  gzclose(fp);
  return pos;
}

// Runs the reference UnionFind over a list of (x,y) merges and reports find(i) for all i.
void ref_unionfind(int n, const int* xs, const int* ys, int m, int* roots_out, int* size_out) {
  UnionFind uf(n);
  for (int i = 0; i < m; i++) uf.merge(xs[i], ys[i]);
  for (int i = 0; i < n; i++) roots_out[i] = uf.find(i);
  *size_out = uf.size();
  uf.clear();
}
}
