#!/bin/bash
# SQ counters of one layer, a few passes (counters only with --kernel-trace: pool rule)
layer=$1; shift
root=$(pwd); out=$root/gpurun_out/pmc_$layer; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM" \
           "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $out/p$i --output-format csv -- python $root/tools/diag/pmc_layer.py $layer "$@" > $out/p$i.log 2>&1 || echo "pass $i failed"
  f=$(ls $out/p$i/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python - $f <<PY
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    acc[r["Kernel_Name"].split("(")[0][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    if "conv_igemm2" in k or "splitk" in k:
        print(k, {c: round(sum(x)/len(x)) for c,x in v.items()}, "n=%d" % len(next(iter(v.values()))))
PY
  rm -rf $out/p$i
done
