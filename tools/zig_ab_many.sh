#!/bin/bash
# A/B of several builds of the generator against the baseline library (tools/attic/zig_bench.py), one box.
for v in "$@"; do echo "=== $v"; python tools/attic/zig_bench.py tools/bin/libbkhip_base.so tools/bin/libbkhip_$v.so 2>&1 | grep -v amdgpu.ids | head -3; done
