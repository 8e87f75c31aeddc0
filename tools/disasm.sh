#!/bin/bash
# Diagnostic: disassemble one kernel of gf_lib.hip (substring of the mangled name) into /tmp/co/<name>.s
set -e
mkdir -p /tmp/co
cd $(dirname "$(dirname "$(readlink -f "$0")")")/goldfish_amd/csrc
hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -c --cuda-device-only --no-gpu-bundle-output $2 gf_lib.hip -o /tmp/co/k.co 2>&1 | grep -E "error" -A3 || true
sym=$(/opt/rocm/lib/llvm/bin/llvm-readelf -s /tmp/co/k.co | grep FUNC | awk '{print $8}' | grep "$1" | head -1)
/opt/rocm/lib/llvm/bin/llvm-objdump -d /tmp/co/k.co --disassemble-symbols=$sym > /tmp/co/$1.s
echo "$sym -> /tmp/co/$1.s ($(wc -l < /tmp/co/$1.s) lines)"
