#!/bin/bash
# Usage: tools/loop_spill_check.sh <path to a rrt_star_v2_body.inc> [extra hipcc flags]
# Compiles rrtx_api.hip for gfx950 with that kernel body (CPU only, ~1 min) and reports, for every streaming loop of
# rppk2t::rrt_star_kernel_v2<true>, the lines per slot, the scratch instructions and the full s_waitcnt vmcnt(0) drains inside it:
# a register-allocation regression of the hot loop shows here before any GPU time is spent.
BODY=$1; shift
D=/tmp/chk_$$; mkdir -p $D; cp /root/repo/robotics-path-planning_amd/csrc/* $D/; cp $BODY $D/rrt_star_v2_body.inc
mkdir -p $D/../include 2>/dev/null
cd $D && sed -i 's#"../../include/rrtx.h"#"/root/repo/include/rrtx.h"#' rrtx_api.hip
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-parentheses-equality -Wno-unused-value "$@" --cuda-device-only -S -Rpass-analysis=kernel-resource-usage -o $D/o.s rrtx_api.hip 2> $D/err.txt
grep -A7 "Function Name: _ZN6rppk2t18rrt_star_kernel_v2ILb1" $D/err.txt | grep -E "VGPRs Spill|ScratchSize|SGPRs Spill" | sed 's/.*remark: *//' | tr '\n' ' '; echo
grep -E "error" $D/err.txt | head -3
L=$(grep -n "^_ZN6rppk2t18rrt_star_kernel_v2ILb1EEEvN4rppk3CtxEi:" $D/o.s | cut -d: -f1)
awk -v l=$L 'NR>=l' $D/o.s | awk '/s_endpgm/{print; exit} {print}' > $D/k.s
python3 - $D/k.s <<'PY'
import sys,re
L=open(sys.argv[1]).read().split('\n')
loads=[i for i,l in enumerate(L) if 'global_load_dwordx4' in l and ' nt' in l]
# group loads: consecutive loads closer than 60 lines = prologue group; loop loads follow
groups=[]; cur=[loads[0]]
for a,b in zip(loads,loads[1:]):
    if b-a<60: cur.append(b)
    else: groups.append(cur); cur=[b]
groups.append(cur)
# a streaming loop = prologue group (>=4 loads) followed by singles
i=0
while i<len(groups):
    g=groups[i]
    if len(g)>=2:
        j=i+1; last=g[-1]
        while j<len(groups) and len(groups[j])==1: last=groups[j][0]; j+=1
        if j>i+1:
            rng=L[g[-1]:last+1]
            sc=[l for l in rng if 'scratch_' in l]
            v0=[l for l in rng if 's_waitcnt vmcnt(0)' in l]
            print("loop lines %d-%d: %d lines/slot, scratch ops %d, vmcnt(0) %d"%(g[-1],last,(last-g[-1])//max(1,(j-i-1)),len(sc),len(v0)))
        i=j
    else: i+=1
PY
rm -rf $D
