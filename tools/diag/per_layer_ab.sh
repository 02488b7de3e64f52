#!/bin/bash
# per-layer eager times of FlowNet2 b4 under two settings of an environment variable: $1=VAR $2=valA $3=valB
env $1=$2 python bench.py --no-extra --no-cpu-baseline --regions 1 --steps 20 --per-layer 2> gpurun_out/pl_a.txt > /dev/null
env $1=$3 python bench.py --no-extra --no-cpu-baseline --regions 1 --steps 20 --per-layer 2> gpurun_out/pl_b.txt > /dev/null
python - <<PY
a=[l.split() for l in open("gpurun_out/pl_a.txt") if " ms " in l]
b=[l.split() for l in open("gpurun_out/pl_b.txt") if " ms " in l]
ta=tb=0
for x,y in zip(a,b):
    fa,fb=float(x[1]),float(y[1]); ta+=fa; tb+=fb
    if abs(fa-fb)>0.002: print("%-52s %8.4f -> %8.4f ms  %7.1f -> %7.1f TF"%(x[0][-52:],fa,fb,float(x[3]),float(y[3])))
print("sum %.3f -> %.3f ms"%(ta,tb))
PY
