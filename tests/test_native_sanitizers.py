"""The host libraries (candidates, annealer, shard reader) under AddressSanitizer + UBSan, and the
multi-worker annealer under ThreadSanitizer for the data that is meant to be race free (the lists
and the fit are mutex-protected; `uses` / `enabled` are atomics).  Host code only: g++, no GPU."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = [os.path.join(HERE, "native", "san_driver.cpp")] + [
    os.path.join(ROOT, "tokenizer_amd", "csrc", f) for f in ("candidates.cpp", "annealer.cpp", "shard_reader.cpp")]


TRIE_SRC = [os.path.join(HERE, "native", "san_trie.cpp"), os.path.join(ROOT, "tokenizer_amd", "csrc", "trie.cpp")]


def _build_and_run(tmp_path, flags, threads, env=None, src=None):
    exe = str(tmp_path / "san_driver")
    subprocess.run(["g++", "-O1", "-g", "-std=c++20", "-pthread", "-fno-omit-frame-pointer"] + flags + ["-o", exe] + (src or SRC),
                   check=True)
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([exe, str(tmp_path), str(threads)], capture_output=True, text=True, env=e, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().endswith("ok")
    assert "runtime error" not in r.stderr and "Sanitizer" not in r.stderr, r.stderr


def test_host_libraries_under_asan_ubsan(tmp_path):
    _build_and_run(tmp_path, ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"], 1,
                   {"ASAN_OPTIONS": "detect_leaks=1"})


def test_trie_builder_under_asan_ubsan(tmp_path):
    # the host side of libtokscore.so: build, hot-first packing, toggles; answers checked against a
    # plain longest match over the token list (duplicates: first id, SolutionFile.cpp:56)
    _build_and_run(tmp_path, ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"], 1,
                   {"ASAN_OPTIONS": "detect_leaks=1"}, src=TRIE_SRC)


def test_multi_worker_annealer_under_tsan(tmp_path):
    # raw_score_ / enabled_cnt_ are read by moves while other workers change them: by design (as in
    # the reference, TokenGenerator.cpp:82-85) and through std::atomic, so TSan has nothing to report
    supp = os.path.join(HERE, "native", "tsan.supp")
    _build_and_run(tmp_path, ["-fsanitize=thread"], 4, {"TSAN_OPTIONS": "suppressions=%s halt_on_error=0" % supp})
