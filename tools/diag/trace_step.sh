set -e
root=$(pwd)
out=$root/gpurun_out/trace_r3
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $out/t --output-format csv -- python $root/bench.py --no-extra --no-cpu-baseline --regions 1 --steps 10 > $out/bench.json 2> $out/err.txt
f=$(ls $out/t/*/*kernel_trace.csv | head -1)
head -1 $f > $out/header.txt
python $root/tools/graph_timeline.py $f -3 -v > $out/timeline.txt
python $root/tools/graph_gaps.py $f >> $out/timeline.txt
python - $f $out/trace_compact.csv <<PY
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
w=csv.writer(open(sys.argv[2],"w"))
w.writerow(["Start_Timestamp","End_Timestamp","Queue_Id","Kernel_Name","Grid_Size_X","Grid_Size_Y","Grid_Size_Z","Workgroup_Size_X"])
for r in rows[-1500:]:
    w.writerow([r["Start_Timestamp"],r["End_Timestamp"],r["Queue_Id"],r["Kernel_Name"].split("(")[0],r["Grid_Size_X"],r["Grid_Size_Y"],r["Grid_Size_Z"],r["Workgroup_Size_X"]])
PY
rm -rf $out/t
tail -30 $out/timeline.txt
