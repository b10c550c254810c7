"""INTEGRATION.md's reference-side binding, compiled: GpuScorer.h is written out of the document as
it stands, built against the reference's own headers and sources (DataFile.cpp / JsonFile.cpp in
place, as oracle/Makefile does) and libtokscore.so, and run.  In this container there is no GPU,
so the stub must end in the library's "no device" error -- never in a CPU result."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
LIB_DIR = os.path.join(ROOT, "tokenizer_amd")

MAIN = r"""
#include <iostream>
#include "files/GpuScorer.h"
#include "files/SolutionFile.h"
int main(int argc, char **argv) {
	DataFile test(argv[2]);
	if (!test.IsValid()) return 3;
	std::vector<std::string> solution = {"a", "b", "ab", "abc"};
	try {
		GpuScore s = ScoreShardOnGpu(solution, test);
		std::cout << s.total_bytes << " characters, " << s.total_tokens << " tokens - compression factor "
		          << (double)s.total_bytes / s.total_tokens << std::endl;
	} catch (const std::runtime_error &e) {
		std::cout << "ERROR " << e.what() << std::endl;
		return 7;
	}
	return 0;
}
"""


def _stub_source():
    text = open(os.path.join(ROOT, "INTEGRATION.md"), encoding="utf-8").read()
    at = text.index("`src/files/GpuScorer.h`")
    m = re.search(r"```cpp\n(.*?)```", text[at:], re.S)
    assert m, "INTEGRATION.md no longer holds the GpuScorer.h block"
    return m.group(1)


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="the reference tree is not present")
def test_gpu_scorer_stub_compiles_against_the_reference_and_fails_loudly_without_a_gpu(tmp_path):
    inc = tmp_path / "files"
    inc.mkdir()
    (inc / "GpuScorer.h").write_text(_stub_source())
    (tmp_path / "main.cpp").write_text(MAIN)
    exe = tmp_path / "stub"
    cmd = ["g++", "-O1", "-std=c++20", "-w", "-I" + str(tmp_path), "-I" + os.path.join(REF, "src"),
           "-I" + os.path.join(REF, "src", "files"), "-I" + os.path.join(REF, "lib", "rapidjson", "include"),
           "-I" + os.path.join(ROOT, "include"), "-o", str(exe), str(tmp_path / "main.cpp"),
           os.path.join(REF, "src", "files", "DataFile.cpp"), os.path.join(REF, "src", "files", "JsonFile.cpp"),
           "-L" + LIB_DIR, "-ltokscore", "-Wl,-rpath," + LIB_DIR, "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    shard = os.path.join(ROOT, "tests", "golden", "dup_tokens", "shard.json")
    r = subprocess.run([str(exe), "unused", shard], capture_output=True, text=True, timeout=120)
    import tokenizer_amd as ta
    if ta.load_library().tokscore_device_count() > 0:
        assert r.returncode == 0 and "compression factor" in r.stdout, (r.stdout, r.stderr)
        assert r.stdout.startswith("15 characters, ")
    else:
        assert r.returncode == 7, (r.returncode, r.stdout, r.stderr)
        assert "ERROR tokscore: " in r.stdout and "no HIP device" in r.stdout
