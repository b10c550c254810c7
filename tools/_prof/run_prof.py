import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import tokenizer_amd as ta
ta.LIB_PATH = os.path.join(ROOT, "tools", "_prof", "libtokscore_prof.so")
from tokenizer_amd import synth
mb = float(sys.argv[1]) if len(sys.argv) > 1 else 100
src, _ = synth.wiki_like(2_000_000, seed=20201120)
tokens = synth.build_vocab(src, 32000)
corpus, off = synth.wiki_like(int(mb * 1e6), seed=20201120)
s = ta.Scorer(0); s.set_vocab(*synth.pack_vocab(tokens)); s.set_corpus(corpus, off)
for _ in range(3): s.score()
print("scan ms", s.last_timing()[1])
lib = ta.load_library()
buf = np.zeros(4096 * 8, dtype=np.uint64)
lib.tokscore_debug_prof.argtypes = [C.c_void_p, C.c_size_t]
print("rc", lib.tokscore_debug_prof(buf.ctypes.data, buf.size))
rows = buf.reshape(4096, 8); rows = rows[rows[:, 6] > 0]
names = ["queue/idle/finish", "tile load+bitmap", "phase1 asm", "flush/limits", "repair asm", "phase2 rest", "total"]
tot = rows[:, 6].astype(float)
print("waves", len(rows), "chunks/wave min/mean/max", rows[:, 7].min(), rows[:, 7].mean(), rows[:, 7].max())
for k, n in enumerate(names):
    v = rows[:, k].astype(float)
    print("%-20s mean %10.0f  (%5.1f %% of mean total)  min %10.0f max %10.0f" % (n, v.mean(), 100 * v.mean() / tot.mean(), v.min(), v.max()))
print("finish time: min %.0f p10 %.0f median %.0f p90 %.0f max %.0f" % tuple(np.percentile(tot, [0, 10, 50, 90, 100])))
# per workgroup: do whole workgroups run slow (placement) or single waves?
allrows = buf.reshape(4096, 8)
wg_chunks = []
for b in range(256):
    r = allrows[b * 16:(b + 1) * 16]
    r = r[r[:, 6] > 0]
    if len(r):
        wg_chunks.append((r[:, 7].sum(), r[:, 6].max(), b))
wg_chunks.sort()
print("chunks per workgroup: min %d p10 %d median %d p90 %d max %d" % tuple(np.percentile([c[0] for c in wg_chunks], [0, 10, 50, 90, 100])))
print("slowest workgroups (chunks, finish, id):", wg_chunks[:8])
print("fastest workgroups:", wg_chunks[-8:])
fin = np.array([c[1] for c in wg_chunks], dtype=float)
print("workgroup finish: min %.0f median %.0f max %.0f" % (fin.min(), np.median(fin), fin.max()))
