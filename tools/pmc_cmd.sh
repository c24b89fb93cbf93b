#!/bin/bash
# tools/pmc_cmd.sh <outdir-tag> "<cmd>" <counter-set> [<counter-set> ...]
# Runs <cmd> once per counter set under rocprofv3 --kernel-trace --pmc (each set = one quoted string of
# counter names) and prints, per dispatch of our kernels, duration and the counters.  GPU box only.
set -u
TAG=$1; CMD=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for SET in "$@"; do
  i=$((i+1))
  # rocprofv3 can hang for minutes in finalisation after a counter-config error: always bound it
  timeout -k 5 ${PMC_TIMEOUT:-150} rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/set$i -o p -- $CMD \
    > $OUT/set$i.log 2>&1 || { echo "set $i ($SET) failed:"; grep -m1 -A1 "error code" $OUT/set$i.log; }
done
python - <<EOF
import csv, collections, glob, os
out="$OUT"
for d in sorted(glob.glob(out+"/set*/")):
    kt={r["Dispatch_Id"]:(int(r["End_Timestamp"])-int(r["Start_Timestamp"])) for r in csv.DictReader(open(d+"p_kernel_trace.csv"))}
    agg=collections.OrderedDict()
    for r in csv.DictReader(open(d+"p_counter_collection.csv")):
        if "ozhip" not in r["Kernel_Name"]: continue
        key=(int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0][-34:])
        agg.setdefault(key,{})[r["Counter_Name"]]=float(r["Counter_Value"])
    print("==",os.path.basename(d.rstrip("/")))
    for (did,k),c in agg.items():
        print("%3d %-34s %8.3f ms  "%(did,k,kt[str(did)]/1e6)+"  ".join("%s=%.4g"%(n,v) for n,v in sorted(c.items())))
EOF
