#!/bin/bash
# A/B of the one-launch rows step (graph-replayed generation steps, us per step): the in-tree library, optionally under several
# settings of an environment knob, against genvc_amd/lib/libgenvc_hip_r05base.so (the round-4 kernels)
# usage: scripts/ab_rows.sh OUTDIR [KNOB "v1 v2 ..."]
out=${1:-gpurun_out/ab}; knob=${2:-GVC_NONE}; vals=${3:-x}; mkdir -p $out
[ -f genvc_amd/lib/libgenvc_hip_r05base.so ] || { echo "build the round-4 sources (git checkout 5b11e86 -- genvc_amd/csrc; python -m genvc_amd.build) and keep the library as genvc_amd/lib/libgenvc_hip_r05base.so first"; exit 1; }
run() { python scripts/time_decode.py $1 $2 $3 2>&1 | grep "us/step (" | tail -1 | sed -E "s/.*decode ([0-9.]+) us.*/\1/"; }
for rep in 1 2 3; do
  for cfg in base $vals; do
    if [ $cfg = base ]; then export GENVC_HIP_LIB=$PWD/genvc_amd/lib/libgenvc_hip_r05base.so; unset $knob; else unset GENVC_HIP_LIB; export $knob=$cfg; fi
    echo "$cfg rep$rep: B8/48-112keys $(run 8 13 64)  B16/48-112 $(run 16 13 64)  B8/110-206 $(run 8 75 96)  B16/110-206 $(run 16 75 96)  B2 $(run 2 13 64)" >> $out/ab.txt
  done
done
unset GENVC_HIP_LIB
cat $out/ab.txt
