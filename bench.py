#!/usr/bin/env python3
"""Headline benchmark: corpus GB/s scored per full-vocabulary evaluation.

A step is ONE full-vocabulary scoring pass (greedy longest-match segmentation of every
document + per-id counts + totals; /root/reference/src/main.cpp:31-46 over
SolutionFile::Tokenize) of the rank's resident shard, followed for N > 1 by one RCCL
all-reduce of counts | total_tokens | total_bytes.  Workload = BASELINE.json configs[1]:
a ~100 MB wiki-like shard (seeded synthetic stand-in) with a 32k
vocabulary per GPU; every rank scores its own shard (weak scaling, no data-path collective
besides the count all-reduce).

    python bench.py [--gpus N --steps K --warmup W] [--mb 100] [--vocab 32000] [--vocab-file tokens.json[.gz]] [--split]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

`--split`: strong scaling -- ONE shard of --mb MB, its documents dealt out to the N ranks in contiguous
blocks (what tests/test_sharding_gloo.py checks on CPU); "scaling": "strong" in the line.

`--gpus N` without a launcher (WORLD_SIZE unset) starts the N ranks itself -- a child
`python -m torch.distributed.run`, before anything here touches the GPU -- and refuses loudly when
fewer than N devices are visible.  `--mb 2500` is configs[2]'s per-GPU share (20 GB over 8 GPUs).

Prints one JSON line on rank 0.  The CPU oracle / compiled reference is timed beside it
(cpu_baseline) on a bounded sample; it is never part of the measured GPU path.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

os.environ["TOKSCORE_TIMING"] = "1"  # the roofline figure needs the library's per-kernel events inside the timed region

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def cpu_baseline(corpus, off, tokens, sample_bytes=100_000_000):  # the whole bench shard: about 10 s of CPU work
    """Reference CPU path on the host cores, on the first ~sample_bytes of the shard."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from tokenizer_amd import synth
    d_end = int(np.searchsorted(off, sample_bytes, side="right")) - 1
    d_end = max(1, min(d_end, len(off) - 1))
    sub_off = off[:d_end + 1]
    nbytes = int(sub_off[-1])
    cores = min(16, os.cpu_count() or 1)  # the GPU box grants 16 host cores per GPU
    out = {}
    ref_result = None
    harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    if os.path.exists(harness):
        with tempfile.TemporaryDirectory() as tmp:
            b = corpus[:nbytes].tobytes()
            texts = [b[int(sub_off[i]):int(sub_off[i + 1])].decode("utf-8") for i in range(d_end)]
            with open(os.path.join(tmp, "t.json"), "w", encoding="utf-8") as f:
                json.dump({"version": "0.4.0", "tokens": [t.decode("utf-8") for t in tokens]}, f, ensure_ascii=False)
            with open(os.path.join(tmp, "s.json"), "w", encoding="utf-8") as f:
                json.dump([{"id": str(i), "title": "", "text": x} for i, x in enumerate(texts)], f, ensure_ascii=False)
            subprocess.run([harness, "score", os.path.join(tmp, "t.json"), os.path.join(tmp, "s.json"),
                            os.path.join(tmp, "e.json"), "--threads", str(cores)], check=True,
                           stdout=subprocess.DEVNULL)
            e = json.load(open(os.path.join(tmp, "e.json")))
        ref_result = e
        out = {"value": nbytes / e["tokenize_seconds"] / 1e9, "unit": "GB/s", "cores": cores, "kind": "reference",
               "sample": "first %d documents (%d bytes) of the rank-0 shard, compiled reference "
                         "SolutionFile::Tokenize on %d threads, Tokenize time only" % (d_end, nbytes, cores)}
    import oracle_ffi as O
    voc = O.OracleVocab(*synth.pack_vocab(tokens))
    t0 = time.time()
    oracle_result = voc.score(corpus[:nbytes], sub_off, threads=cores)
    port = nbytes / (time.time() - t0) / 1e9
    if not out:
        out = {"value": port, "unit": "GB/s", "cores": cores, "kind": "port",
               "sample": "first %d documents (%d bytes) of the rank-0 shard, oracle/tokscore_oracle.c on %d threads"
                         % (d_end, nbytes, cores)}
    out["port_value"] = port
    return out, {"bytes": nbytes, "docs": d_end, "oracle": oracle_result, "reference": ref_result}


def load_vocab_file(path):
    """tokens.json[.gz] as the reference's TokenGenerator writes it (SolutionFile.cpp:27-41): {"tokens": [...]}"""
    import gzip
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "rt", encoding="utf-8") as f:
        return [t.encode("utf-8") for t in json.load(f)["tokens"]]


def split_shard(corpus, off, rank, world):
    """--split: rank r's contiguous block of the shard's documents, blocks of (nearly) equal byte size -- the
    partition tests/test_sharding_gloo.py checks (tokenizer_amd/shard.py)."""
    from tokenizer_amd import shard
    return shard.shard_corpus(corpus, off, rank, world)


def settle(scorer, out_ptr, stream_ptr, sync):
    """Before any timing: the evaluations in which the library still arranges itself -- the first one repairs
    every range's entry state, and the automaton scan lays LDS out again from its own counters after the
    first one (tokscore_info.dfa_relayouts).  -> milliseconds of each evaluation, host clock."""
    ms = []
    for _ in range(7):
        t0 = time.perf_counter()
        scorer.score_async(out_ptr, stream_ptr)
        sync()
        ms.append((time.perf_counter() - t0) * 1e3)
        if len(ms) >= 3 and abs(ms[-1] - ms[-2]) < 0.3 * ms[-2]:  # (steady: no re-layout, no repairs in the last two)
            break
    return ms


def spin_up(scorer, out_ptr, stream_ptr, sync, block=50, max_blocks=6):
    """Untimed, before the warm-up steps: evaluations back to back in blocks of 50 until a block is no faster than the one
    before it (1 %).  An idle MI355X takes some ten milliseconds of work to reach the clocks it then holds: the first
    hundred 0.1 ms evaluations after a pause run 3 - 5 % slower than the next hundred (tools/dbg/ab_lib.py), and the
    timed region is to measure the state a scoring loop is in, not the ramp.  -> ms per evaluation of each block."""
    per = []
    for _ in range(max_blocks):
        sync()
        t0 = time.perf_counter()
        for _ in range(block):
            scorer.score_async(out_ptr, stream_ptr)
        sync()
        per.append((time.perf_counter() - t0) * 1e3 / block)
        if len(per) >= 2 and per[-1] > 0.99 * per[-2]:
            break
    return per


def after_toggle(scorer, tokens, out_ptr, stream_ptr, sync):
    """What one toggle costs the next full evaluations (host clock, ms): the first ones run on the plan's own
    scan, then the automaton's image is rebuilt from the vocabulary as it is."""
    multi = [i for i, t in enumerate(tokens) if len(t) > 3]
    scorer.toggle(int(multi[len(multi) // 2]), False)
    ms, active = [], []
    for _ in range(6):
        t0 = time.perf_counter()
        scorer.score_async(out_ptr, stream_ptr)
        sync()
        ms.append((time.perf_counter() - t0) * 1e3)
        active.append(int(scorer.info().dfa_active))
    scorer.toggle(int(multi[len(multi) // 2]), True)
    settle(scorer, out_ptr, stream_ptr, sync)
    return {"ms": ms, "automaton_scan": active, "note": "full evaluations after ONE toggle, host clock with a sync each"}


def parity_check(gpu_vec, V, total_bytes, checked):
    """configs[1]'s "bit-exact count check", in the run: the GPU's counts | total_tokens |
    total_bytes for the whole shard against what the CPU side of cpu_baseline produced for the same
    bytes -- the compiled reference (oracle/_ref, main.cpp:38-45 over SolutionFile::Tokenize) when
    present, and the C restatement.  Only when the CPU sample was the whole shard."""
    if checked["bytes"] != total_bytes:
        return {"against": None, "bytes": checked["bytes"], "exact": None,
                "note": "the CPU sample is not the whole shard: nothing to compare"}
    counts, tt, tb = checked["oracle"]
    exact = bool(np.array_equal(gpu_vec[:V + 3], counts)) and int(gpu_vec[V + 3]) == int(tt) and int(gpu_vec[V + 4]) == int(tb)
    against = "oracle"
    ref = checked["reference"]
    if ref is not None:
        rc = np.zeros(V + 3, dtype=np.uint64)
        for k, v in ref["counts"].items():
            rc[V + 2 if int(k) < 0 else int(k)] = v
        exact = exact and bool(np.array_equal(gpu_vec[:V + 3], rc)) and int(gpu_vec[V + 3]) == int(ref["total_tokens"]) \
            and int(gpu_vec[V + 4]) == int(ref["total_bytes"])
        against = "reference"
    return {"against": against, "bytes": checked["bytes"], "docs": checked["docs"], "exact": exact}


# ---- the step and its reduction: what tests/test_sharding_gloo.py drives on CPU with gloo ----

def reduce_result(out, dist, world, host_reduce=False):
    """The one exchange of the path: in-place SUM all-reduce of counts | total_tokens | total_bytes
    (int64 view of the u64 words: wrap-around addition).  host_reduce: through a host copy (gloo)."""
    if world <= 1:
        return out
    if host_reduce:
        host_copy = out.cpu()
        dist.all_reduce(host_copy)
        out.copy_(host_copy)
    else:
        dist.all_reduce(out)
    return out


def timed_steps(step, fence, steps, warmup, after_warmup=None):
    """W untimed steps, then exactly K steps between two fences; seconds of the K steps."""
    for _ in range(warmup):
        step()
    fence()
    if after_warmup:
        after_warmup()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    return time.perf_counter() - t0


def max_over_ranks(elapsed, dist, world, device):
    import torch
    if world <= 1:
        return elapsed
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def launch_ranks(args):
    """--gpus N without a launcher: start the N ranks (one per GPU, RCCL) as a child process, before
    this process has touched a GPU; its exit code is ours."""
    import torch  # (device_count does not initialise the GPU)
    have = torch.cuda.device_count()
    if have < args.gpus:
        sys.stderr.write("bench.py: --gpus %d asked for, %d GPU(s) visible: refusing to run fewer ranks under that label\n"
                         % (args.gpus, have))
        return 2
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def measured_traffic(workload_bytes):
    """HBM bytes per scan launch from the newest committed rocprofv3 --pmc summary of this same
    command (tools/refresh_profiles.sh + tools/summarize_profiles.py; FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950, plus WRITE_SIZE).  None when there is none or it
    was taken on another shard size."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")))
    for f in reversed(files):
        try:
            d = json.load(open(f))
            bench = json.loads(open(f.replace("_pmc.json", "_bench.json")).read().strip().splitlines()[-1])
        except (OSError, ValueError):
            continue
        if "traffic_bytes_per_launch" in d and bench.get("config", {}).get("bytes_per_gpu") == workload_bytes:
            return float(d["traffic_bytes_per_launch"]), os.path.relpath(f, ROOT)
    return None, None


def moves_per_sec(scorer, tokens, corpus, off):
    """BASELINE.json's second figure, outside the timed region: (ii) toggle one
    token + incremental re-score on the resident shard, against a full re-score per move; (i) the
    host annealer's own loop (libtokanneal.so) on the candidates of a 1 MB slice."""
    import os
    from tokenizer_amd import annealing
    rng = np.random.default_rng(5)
    multi = [i for i, t in enumerate(tokens) if len(t.decode("utf-8", "ignore")) > 1]
    picks = rng.choice(multi, size=120)
    buf = np.zeros(len(tokens) + 3, dtype=np.uint64)
    scorer.score(out=buf)
    on = np.ones(len(tokens), dtype=bool)

    def run(incremental, which):
        t0 = time.perf_counter()
        for i in which:
            on[i] = not on[i]
            scorer.toggle(int(i), bool(on[i]))
            scorer.score(incremental=incremental, out=buf)
        return len(which) / (time.perf_counter() - t0)

    # (iii) K = 64 independent moves judged by one tokscore_score_toggles call, none applied
    small = len(corpus) <= 500_000_000  # (calls of a fraction of a millisecond: enough of them that the first ones' share is small)
    batches = [rng.choice(multi, size=64, replace=False) for _ in range(25 if small else 6)]
    for _ in range(3 if small else 1):  # (the first call's buffers; the occurrence index where the one-pass search does not serve)
        scorer.score_toggles(batches[0], 0)
    t0 = time.perf_counter()
    deltas = [scorer.score_toggles(b, 0) for b in batches[1:]]
    batched = 64 * len(deltas) / (time.perf_counter() - t0)
    # ... and 256 per call: one pass of the search automaton over the corpus costs the same for any number of moves
    big = [rng.choice(multi, size=256, replace=False) for _ in range(13 if small else 4)]
    scorer.score_toggles(big[0], 0)
    t0 = time.perf_counter()
    big_deltas = [scorer.score_toggles(b, 0) for b in big[1:]]
    batched256 = 256 * len(big_deltas) / (time.perf_counter() - t0)
    search_passes = int(scorer.info().search_passes)
    base_tt = scorer.score(out=buf)[1]
    batched_exact = True
    for k in (0, 21, 42, 63):  # against two full evaluations
        scorer.toggle(int(batches[1][k]), False)
        tt = scorer.score(out=buf)[1]
        scorer.toggle(int(batches[1][k]), True)
        batched_exact = batched_exact and int(deltas[0][k]) == tt - base_tt
    for k in (5, 250):
        scorer.toggle(int(big[1][k]), False)
        tt = scorer.score(out=buf)[1]
        scorer.toggle(int(big[1][k]), True)
        batched_exact = batched_exact and int(big_deltas[0][k]) == tt - base_tt
    scorer.score(out=buf)
    # (iv) the same call inside an annealing-like loop: 64 moves judged, the best one applied, the exact
    # score taken, again -- every call then starts from a changed vocabulary
    state = np.ones(len(tokens), dtype=bool)
    loop_batches = [rng.choice(multi, size=64, replace=False) for _ in range(21)]
    t0 = 0.0
    for n, b in enumerate(loop_batches):
        if n == 1:
            t0 = time.perf_counter()  # (the first batch is the warm-up)
        en = [0 if state[i] else 1 for i in b]
        d = scorer.score_toggles(b, en)
        k = int(np.argmin(d))
        state[b[k]] = bool(en[k])
        scorer.toggle(int(b[k]), bool(en[k]))
        scorer.score(out=buf)
    anneal_like = 64 * (len(loop_batches) - 1) / (time.perf_counter() - t0)
    for i in np.nonzero(~state)[0]:  # back to the full vocabulary
        scorer.toggle(int(i), True)
    scorer.score(out=buf)
    run(True, picks[:2])
    inc = run(True, picks[2:])
    full = run(False, picks[2:42])
    docs = int(np.searchsorted(off, 1_000_000))
    cands = annealing.GetTokens(corpus[:int(off[docs])], off[:docs + 1], 10)
    n_cand = len(cands)
    threads = min(16, os.cpu_count() or 1)
    gen = annealing.TokenGenerator(cands, 30000, max(threads, 16))
    t0 = time.perf_counter()
    st = gen.Generate(0, threads=threads, seed=5)[-1]
    host = st.gen_cnt / (time.perf_counter() - t0)
    gen.close()
    return {"gpu_score_toggles_k64": batched, "gpu_score_toggles_k256": batched256, "gpu_score_toggles_exact": batched_exact,
            "gpu_score_toggles_by": "one pass of a search automaton per 256 moves" if search_passes else "occurrence index + search per (move, chunk)",
            "gpu_anneal_like_loop_k64": anneal_like,
            "gpu_toggle_incremental_rescore": inc, "gpu_toggle_full_rescore": full,
            "host_dag_loop": host, "host_threads": threads, "host_candidates": n_cand,
            "note": "moves/s; GPU: this shard and vocabulary, result on the host after every move"}


def two_in_flight(first, tokens, corpus, off, dev, steps):
    """Outside the timed region and not the headline: the same shard scored for two candidate
    vocabularies at a time, one context and one stream each, so that one evaluation's histogram
    and the tail of its scan overlap the other's scan (what an annealer that evaluates two
    candidates concurrently gets).  Both results are checked against the synchronous call."""
    import torch
    import tokenizer_amd as ta
    from tokenizer_amd import synth
    second = ta.Scorer(dev.index or 0)
    try:
        second.set_vocab(*synth.pack_vocab(tokens))
        second.set_corpus(corpus, off)
        second.toggle(100, False)  # the other candidate: one token fewer
        scorers = [first, second]
        ref = [s.score() for s in scorers]
        outs = [torch.zeros(len(tokens) + 5, dtype=torch.int64, device=dev) for _ in scorers]
        streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
        dt = 0.0
        for n in (3, steps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                for k in range(2):
                    scorers[k].score_async(outs[k].data_ptr(), streams[k].cuda_stream)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        exact = True
        for k in range(2):
            got = outs[k].cpu().numpy().view(np.uint64)
            exact = exact and bool(np.array_equal(got[:len(tokens) + 3], ref[k][0])) and int(got[len(tokens) + 3]) == ref[k][1]
        return {"value": len(corpus) * 2 * steps / dt / 1e9, "unit": "GB/s", "ms_per_evaluation": dt * 1e3 / (2 * steps),
                "evaluations": 2 * steps, "exact": exact,
                "note": "two contexts on two streams, same shard; not the headline (one evaluation per step)"}
    finally:
        second.close()


def annealed_vocab_block(dev, args, synth, ta):
    """Not the headline: the same measurement with the vocabulary the REFERENCE's own annealer wrote
    (tests/golden/wiki_annealed/tokens.json.gz: 30,275 tokens, 124,789 trie nodes, most prefixes not tokens) on
    --mb MB of the kind of text it was annealed on (synth.wiki_like seed 1234, 20,000 words), checked against the
    C restatement on the whole shard.  A vocabulary of that shape does not suit the flattened automaton (2.8 M virtual
    entries); the library takes the automaton's failure form for it (dfa.h: DfaImage::fail_code, dfa_fscan_kernel)."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_ffi as O
    path = os.path.join(ROOT, "tests", "golden", "wiki_annealed", "tokens.json.gz")
    tokens = load_vocab_file(path)
    corpus, off = synth.wiki_like(int(args.mb * 1e6), seed=1234, n_words=20_000)
    V = len(tokens)
    s = ta.Scorer(dev.index or 0)
    try:
        s.set_vocab(*synth.pack_vocab(tokens))
        s.set_corpus(corpus, off)
        out = torch.zeros(V + 5, dtype=torch.int64, device=dev)
        stream = torch.cuda.current_stream()
        settle(s, out.data_ptr(), stream.cuda_stream, torch.cuda.synchronize)
        spin_up(s, out.data_ptr(), stream.cuda_stream, torch.cuda.synchronize, block=20, max_blocks=4)
        s.set_timing(1)
        s.drain_timings()
        n = max(3, args.steps // 2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            s.score_async(out.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        tm = np.array(s.drain_timings())
        got = out.cpu().numpy().view(np.uint64)
        info = s.info()
        voc = O.OracleVocab(*synth.pack_vocab(tokens))
        counts, tt, tb = voc.score(corpus, off, threads=min(16, os.cpu_count() or 1))
        exact = bool(np.array_equal(got[:V + 3], counts)) and int(got[V + 3]) == int(tt) and int(got[V + 4]) == int(tb)
        scan_ms = float(tm[:, 1].mean())
        alg = int(info.corpus_bytes) + 8 * (V + 3)
        return {"value": len(corpus) * n / dt / 1e9, "unit": "GB/s", "ms_per_step": dt / n * 1e3, "steps": n,
                "vocab": V, "trie_nodes": int(info.trie_nodes), "bytes": int(info.corpus_bytes), "total_tokens": int(got[V + 3]),
                "scan_path": ("lds_automaton_failure_form" if info.dfa_form == 2 else "lds_automaton") if info.dfa_active
                             else {1: "lds_tile", 2: "lds_stream"}.get(int(info.scan_path), "generic"),
                "roofline": {"kernel_ms": scan_ms, "achieved": alg / (scan_ms * 1e-3) / 1e9, "frac": alg / (scan_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                             "eval_kernels_ms": float(tm[:, 0].mean())},
                "parity_checked": {"against": "oracle", "bytes": int(info.corpus_bytes), "exact": exact},
                "workload": "tests/golden/wiki_annealed/tokens.json.gz on wiki-like %.0f MB (seed 1234, 20,000 words)" % args.mb}
    finally:
        s.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mb", type=float, default=100.0, help="shard size per GPU in MB")
    ap.add_argument("--vocab", type=int, default=32000)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--vocab-file", default=None, help="tokens.json[.gz] of the reference's format instead of the built vocabulary")
    ap.add_argument("--split", action="store_true", help="strong scaling: one --mb shard, its documents split over the ranks")
    ap.add_argument("--seed", type=int, default=20201120)
    ap.add_argument("--words", type=int, default=0, help="lexicon size of the synthetic text (0: synth.wiki_like's default)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and os.environ.get("BENCH_REHEARSAL") != "1":
        sys.exit(launch_ranks(args))

    import torch
    import tokenizer_amd as ta
    from tokenizer_amd import synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # BENCH_REHEARSAL=1: all ranks on GPU 0 over gloo -- exercises the N > 1 code path on a
    # one-GPU box (RCCL refuses two ranks on one device); never used for a reported number.
    rehearsal = os.environ.get("BENCH_REHEARSAL") == "1"
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but the launcher started %d rank(s)\n" % (args.gpus, world))
        sys.exit(2)
    if rehearsal:
        local_rank = 0
    elif torch.cuda.device_count() <= local_rank:
        sys.stderr.write("bench.py: rank %d has no GPU (%d visible)\n" % (rank, torch.cuda.device_count()))
        sys.exit(2)
    dist = None
    # BENCH_RCCL_SELFTEST=1 (one rank): run the exchange of the N > 1 path anyway -- an RCCL
    # communicator of one rank, the same all_reduce calls on the same stream after score_async -- so
    # that the collective code has run on hardware at all; labelled in the line, never the headline.
    selftest = world == 1 and os.environ.get("BENCH_RCCL_SELFTEST") == "1"
    if world > 1 or selftest:
        import torch.distributed as dist
        if selftest:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    ranks_in_exchange = 2 if selftest else world  # (> 1: reduce_result / max_over_ranks do call the collective)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    # ---- synthetic shard (configs[1] stand-in) and a vocabulary shared by all ranks ----
    nbytes = int(args.mb * 1e6)
    kw = {"n_words": args.words} if args.words else {}
    if args.vocab_file:
        tokens = load_vocab_file(args.vocab_file)
    else:
        vocab_src, _ = synth.wiki_like(2_000_000, seed=args.seed, **kw)
        tokens = synth.build_vocab(vocab_src, args.vocab)
    if args.split:  # one shard for the whole job, this rank's block of its documents
        corpus, off = synth.wiki_like(nbytes, seed=args.seed, **kw)
        corpus, off = split_shard(corpus, off, rank, world)
    else:
        corpus, off = synth.wiki_like(nbytes, seed=args.seed + rank, **kw)
    V = len(tokens)

    scorer = ta.Scorer(local_rank)  # raises when the HIP library or the GPU is missing
    scorer.set_vocab(*synth.pack_vocab(tokens))
    t0 = time.time()
    scorer.set_corpus(corpus, off)
    h2d_s = time.time() - t0

    out = torch.zeros(V + 5, dtype=torch.int64, device=dev)  # counts | total_tokens | total_bytes
    stream = torch.cuda.current_stream()

    dist_mod = dist if ranks_in_exchange > 1 else None

    def step():
        scorer.score_async(out.data_ptr(), stream.cuda_stream)
        reduce_result(out, dist_mod, ranks_in_exchange, host_reduce=rehearsal)

    def fence():
        torch.cuda.synchronize()
        if ranks_in_exchange > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # HIP events, set by the library on the stream its kernels run on and read after the timed
    # region (no host sync inside it): on the scan kernel of every 4th timed launch (the stream scan's
    # dispatch carries them itself; the other scan paths have a record before and after).  Marking
    # every launch, and every kernel of it, costs 8 % of a step (a record holds the stream for
    # 3 - 4 us; tools/dbg/ab_steps.py), which a caller without a profiler does not pay.
    def sync():
        torch.cuda.synchronize()

    settle_ms = settle(scorer, out.data_ptr(), stream.cuda_stream, sync)
    spin_ms = spin_up(scorer, out.data_ptr(), stream.cuda_stream, sync)
    sample_every = 4 if args.steps >= 8 else 1
    scorer.set_timing(2, sample_every)
    elapsed = timed_steps(step, fence, args.steps, args.warmup, after_warmup=scorer.drain_timings)
    scan_ms = [t[1] for t in scorer.drain_timings()]
    # the whole evaluation's kernels, every one marked: five launches after the timed region
    scorer.set_timing(1)
    for _ in range(5):
        scorer.score_async(out.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize()
    eval_ms = [t[0] for t in scorer.drain_timings()]
    elapsed = max_over_ranks(elapsed, dist_mod, ranks_in_exchange, "cpu" if rehearsal else dev)
    # what d_out cannot show: a token-stream overflow of the last asynchronous evaluation
    overflowed = scorer.async_status(stream.cuda_stream)
    if overflowed:
        sys.stderr.write("bench.py: rank %d: the asynchronous evaluation overflowed its token stream; result invalid\n" % rank)
        sys.exit(3)

    host = out.cpu().numpy().view(np.uint64)
    total_bytes_all = int(host[V + 4])  # after the all-reduce: all ranks' bytes
    if world == 1:
        local_vec = host.copy()
    else:  # this rank's own evaluation, for the parity check (the all-reduced vector is the sum over ranks)
        scorer.score_async(out.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize()
        local_vec = out.cpu().numpy().view(np.uint64).copy()
    info = scorer.info()
    parity_ranks = None
    if world > 1 and not args.no_cpu:
        # configs[1]'s bit-exact count check at N > 1: every rank's own evaluation against the C restatement of the
        # reference's path on the same shard (16 host threads per rank), the verdicts gathered with a MIN all-reduce
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_ffi as O
        voc = O.OracleVocab(*synth.pack_vocab(tokens))
        counts, tt, tb = voc.score(corpus, off, threads=min(16, os.cpu_count() or 1))
        mine = bool(np.array_equal(local_vec[:V + 3], counts)) and int(local_vec[V + 3]) == int(tt) and int(local_vec[V + 4]) == int(tb)
        flag = torch.tensor([1 if mine else 0], dtype=torch.int64, device="cpu" if rehearsal else dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        parity_ranks = {"against": "oracle", "ranks": world, "exact": bool(int(flag.item()) == 1),
                        "note": "every rank's own shard against oracle/tokscore_oracle.c; MIN over ranks"}
    if rank == 0:
        value = total_bytes_all * args.steps / elapsed / 1e9
        alg_bytes = int(info.corpus_bytes) + 8 * (V + 3)
        if info.dfa_active and info.dfa_form == 2:  # (--vocab-file with a vocabulary whose prefixes are mostly not tokens)
            scan_kernel, scan_name = "dfa_fscan_kernel<%d, false>" % info.dfa_chains, "lds_automaton_failure_form"
        elif info.dfa_active:
            scan_kernel, scan_name = "dfa_scan_asm_kernel<%d>" % info.dfa_chains, "lds_automaton"
        else:
            scan_kernel = {1: "scan_tile_kernel", 2: "scan_ranges_kernel"}.get(int(info.scan_path), "scan_generic_kernel")
            scan_name = {1: "lds_tile", 2: "lds_stream"}.get(int(info.scan_path), "generic")
        scan_avg_ms = float(np.mean(scan_ms))
        achieved = alg_bytes / (scan_avg_ms * 1e-3) / 1e9
        traffic, traffic_src = measured_traffic(int(info.corpus_bytes))
        # what a kernel that only reads gets out of this box: over 4 GiB (HBM)
        # and over a buffer of the shard's size (which the 256 MB last-level cache holds)
        stream_hbm = scorer.probe_read_bandwidth(4 << 30, 5)
        stream_shard = scorer.probe_read_bandwidth(max(1 << 20, int(info.corpus_bytes)), 20)
        rec = {
            "metric": "corpus GB/s scored per full-vocab eval",
            "value": value, "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if args.split else "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic" + (" (REHEARSAL: ranks share GPU 0, gloo)" if rehearsal else "")
            + (" (RCCL SELFTEST: one rank, the all-reduce of the N > 1 path inside every step)" if selftest else ""),
            "config": {"workload": ("configs[1]: ONE wiki-like %.0f MB shard (seed %d) split by documents over the ranks, "
                                    "one full-vocabulary scoring pass, V=%d" if args.split else
                                    "configs[1]: wiki-like %.0f MB shard per GPU (seed %d+rank), "
                                    "one full-vocabulary scoring pass, V=%d") % (args.mb, args.seed, V)
                                   + (", vocabulary from %s" % os.path.basename(args.vocab_file) if args.vocab_file else ""),
                       "docs_per_gpu": int(info.corpus_docs), "bytes_per_gpu": int(info.corpus_bytes),
                       "vocab": V, "max_token_bytes": int(info.max_len), "trie_nodes": int(info.trie_nodes), "trie_slots": int(info.trie_slots),
                       "trie_lds_slots": int(info.trie_lds_slots), "chunks": int(info.chunks),
                       "scan_path": scan_name,
                       "automaton": {"slots": int(info.dfa_slots), "virtual_slots": int(info.dfa_virtual_slots), "lds_slots": int(info.dfa_lds_slots),
                                     "lds_used": int(info.dfa_lds_used), "chains": int(info.dfa_chains), "relayouts": int(info.dfa_relayouts),
                                     "lookups_answered_by_L2_per_evaluation": int(info.dfa_second_hits)} if info.dfa_active else None,
                       "parallelism": "documents sharded over %d GPU(s), 1 count all-reduce" % world},
            "roofline": {"bound": "hbm", "kernel": scan_kernel,
                         "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": traffic, "traffic_source": traffic_src, "alg_bytes_per_launch": alg_bytes, "kernel_ms": scan_avg_ms,
                         "kernel_ms_from": "HIP events attached to the kernel's dispatch (hipExtLaunchKernel: its own begin and end) on %d of the %d timed launches (every %s)"
                                           % (len(scan_ms), args.steps, "one" if sample_every == 1 else "%dth" % sample_every),
                         "eval_kernels_ms": float(np.mean(eval_ms)),
                         "eval_kernels_ms_from": "5 launches after the timed region, every kernel marked",
                         "measured_stream_read_GBps": {"hbm_4GiB": stream_hbm, "shard_sized_buffer": stream_shard},
                         "frac_of_measured_hbm_stream": achieved / stream_hbm if stream_hbm else None},
            "total_tokens": int(host[V + 3]), "h2d_inclusive_GBps": info.corpus_bytes / (h2d_s + elapsed / args.steps) / 1e9,
            # once per corpus and vocabulary, outside the timed region: the plan (tokscore_set_corpus: upload, chunking,
            # stream image), the automaton's image and buffers, and the evaluations that still arrange things
            "plan_ms": {"set_corpus_host_ms": h2d_s * 1e3, "automaton_plan_ms": float(info.dfa_plan_ms), "automaton_image_build_ms": float(info.dfa_build_ms)},
            "first_evaluation_ms": settle_ms[0], "settling_evaluations_ms": settle_ms,
            "spin_up": {"evaluations": 50 * len(spin_ms), "ms_per_evaluation_by_block_of_50": spin_ms,
                        "note": "untimed, before the warm-up steps: until the device holds its clocks (a block no faster than the one before)"},
        }
        if world == 1 and not args.no_cpu:
            rec["two_evaluations_in_flight"] = two_in_flight(scorer, tokens, corpus, off, dev, args.steps)
            rec["cpu_baseline"], checked = cpu_baseline(corpus, off, tokens)
            rec["parity_checked"] = parity_check(local_vec, V, int(info.corpus_bytes), checked)
            rec["eval_after_toggle_ms"] = after_toggle(scorer, tokens, out.data_ptr(), stream.cuda_stream, sync)
            rec["annealing_moves_per_s"] = moves_per_sec(scorer, tokens, corpus, off)
            if not args.vocab_file:
                rec["headline_annealed_vocab"] = annealed_vocab_block(dev, args, synth, ta)
        if parity_ranks is not None:
            rec["parity_checked"] = parity_ranks
        print(json.dumps(rec))
        if rec.get("parity_checked", {}).get("exact") is False:
            sys.stderr.write("bench.py: the GPU result differs from the CPU %s on the bench shard\n" % rec["parity_checked"]["against"])
            scorer.close()
            sys.exit(4)
    scorer.close()
    if ranks_in_exchange > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
