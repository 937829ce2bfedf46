#!/bin/bash
# A/B of the one-stream one-launch step under settings of an environment knob (graph-replayed generation steps, us per step)
# usage: scripts/ab_persist.sh OUTDIR KNOB "v1 v2 ..."
out=${1:-gpurun_out/abp}; knob=$2; vals=$3; mkdir -p $out
run() { python scripts/time_decode.py 1 $1 $2 2>&1 | grep "us/step (" | tail -1 | sed -E "s/.*decode ([0-9.]+) us.*/\1/"; }
for rep in 1 2 3; do
  for v in $vals; do
    export $knob=$v
    echo "$knob=$v rep$rep: 48-112 keys $(run 13 64)  110-206 keys $(run 75 96)" >> $out/ab_$knob.txt
  done
done
cat $out/ab_$knob.txt
