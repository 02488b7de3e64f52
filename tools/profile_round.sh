#!/bin/bash
# Round-end measurement on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r01
# Writes rocprofv3 kernel statistics, the bench JSON lines and the PMC traffic summaries under gpurun_out/profiles_<tag>/
# (copy what should be judged into profiles/).  Counters are collected in their own passes (no trace domains mixed in).
set -e
tag=${1:-r03}
root=$(pwd)
out=$root/gpurun_out/profiles_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp

stats() {  # name, bench args...
  local name=$1; shift
  rm -rf $out/tmp_$name
  rocprofv3 --kernel-trace --stats -d $out/tmp_$name --output-format csv -- python $root/bench.py "$@" --no-cpu-baseline \
      > $out/${tag}_${name}_bench_under_rocprof.json 2> $out/tmp_$name.err
  cp $(ls $out/tmp_$name/*/*kernel_stats.csv | head -1) $out/${tag}_${name}_kernel_stats.csv
  rm -rf $out/tmp_$name
  echo "stats $name done"
}

# the driver's own command line (FlowNet2 b4, BASELINE config 3) -- without the FlowNetC extra, so that the per-kernel
# averages of the summary are those of ONE workload
stats flownet2_b4_f16x2 --no-extra
stats flownetc_b8_f16x2 --model FlowNetC --batch 8
stats flownets_b8_f16x2 --model FlowNetS --batch 8 --no-extra
stats flownets_train_b8_f16x2 --mode train --steps 10 --warmup 3

pmc() {  # name, counter, bench args...
  local name=$1 counter=$2; shift 2
  rm -rf $out/pmc_${name}_$counter
  rocprofv3 --kernel-trace --pmc $counter -d $out/pmc_${name}_$counter --output-format csv -- python $root/bench.py "$@" \
      --no-cpu-baseline --no-graph --steps 3 --warmup 1 > /dev/null 2> $out/pmc_${name}_$counter.err
}
export FN2_TRAIN_GRAPH=0   # (counter passes name kernels per launch: the eager train step)
for cfg in "FlowNet2_b4_f16x2 --model FlowNet2 --batch 4 --dtype f16x2 --no-extra --regions 1" "FlowNetC_b8_f16x2 --model FlowNetC --batch 8 --dtype f16x2 --no-extra --regions 1" "FlowNetS_train_b8_f16x2 --mode train"; do
  set -- $cfg
  name=$1; shift
  pmc $name FETCH_SIZE "$@"
  pmc $name WRITE_SIZE "$@"
  python $root/tools/pmc_traffic.py $(ls $out/pmc_${name}_FETCH_SIZE/*/*counter_collection.csv | head -1) \
      $(ls $out/pmc_${name}_WRITE_SIZE/*/*counter_collection.csv | head -1) > $out/pmc_traffic_$name.json
  rm -rf $out/pmc_${name}_FETCH_SIZE $out/pmc_${name}_WRITE_SIZE
  echo "pmc $name done"
done
unset FN2_TRAIN_GRAPH

python $root/tools/bench_ops.py > $out/${tag}_ops_bandwidth_b8.json 2> /dev/null
python $root/tools/bench_ops.py --batch 64 > $out/${tag}_ops_bandwidth_b64.json 2> /dev/null
# the plain (un-profiled) headline run: the driver's command line, CPU baseline and the FlowNetC extra included
python $root/bench.py > $out/${tag}_default_flownet2_b4_f16x2_bench.json 2> /dev/null
python $root/bench.py --height 436 --width 1024 --steps 10 --warmup 3 --no-cpu-baseline \
    > $out/${tag}_flownet2_b4_1024x436_f16x2_bench.json 2> /dev/null
python $root/bench.py --model FlowNetC --batch 8 --no-extra --no-cpu-baseline > $out/${tag}_flownetc_b8_f16x2_bench.json 2> /dev/null
python $root/bench.py --model FlowNetS --batch 8 --no-extra --no-cpu-baseline > $out/${tag}_flownets_b8_f16x2_bench.json 2> /dev/null
FN2_DIST_SINGLE=1 python $root/bench.py --no-cpu-baseline > $out/${tag}_default_rccl_single_rank_rehearsal.json 2> /dev/null
python $root/bench.py --gpus 2 --no-cpu-baseline > $out/${tag}_default_gpus2_gloo_rehearsal.json 2> /dev/null
python $root/bench.py --mode train --steps 10 --warmup 3 > $out/${tag}_flownets_train_b8_f16x2_bench.json 2> /dev/null
ls -la $out
