#!/bin/bash
# the scan's time against the shard's size: what a launch costs before its first step and after its last
for mb in 25 50 100 200 400; do timeout -k 10 200 python3 tools/dbg/ab_lib.py 100 $mb | tail -1 | sed "s/^/$mb MB: /" || exit 1; done
