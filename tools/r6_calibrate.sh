#!/bin/bash
# ON THE GPU BOX: the two models under bench.py's roofline section, measured.
#   (b) issue slots: tools/felab/issuelab issue              -> gpurun_out/r6_cal/issue_slots.txt      (-> profiles/r6_issue_slots.txt)
#   (a) FETCH_SIZE against a known byte count in three access patterns (wide streaming reads, 64-byte row segments at the row strides of
#       the first transform pass at 2^20 and 2^22): rocprofv3 --pmc FETCH_SIZE over `issuelab fetch <pattern>`, one counter per pass
#                                                            -> gpurun_out/r6_cal/fetch_factor.txt     (-> profiles/r6_fetch_factor.txt)
set -u
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r6_cal; rm -rf $O; mkdir -p $O
B=$R/tools/felab/_build/issuelab
[ -x $B ] || { mkdir -p $R/tools/felab/_build && hipcc -O3 -std=c++17 --offload-arch=gfx950 $R/tools/felab/issuelab.hip -o $B; }
$B issue > $O/issue_slots.txt 2>&1
cd /tmp
for p in wide seg16k seg32k; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch_$p -o p -- $B fetch $p > $O/fetch_$p.log 2>&1
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d $O/rdreq_$p -o p -- $B fetch $p > $O/rdreq_$p.log 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $O/tcc_$p -o p -- $B fetch $p > $O/tcc_$p.log 2>&1
done
cd $R
python - <<'PY' > $O/fetch_factor.txt
import csv, glob, os
O = os.path.join(os.getcwd(), "gpurun_out", "r6_cal")
BYTES = 1 << 30
print("FETCH_SIZE (rocprofv3 --pmc, gfx950) against a known byte count: every launch of tools/felab/issuelab.hip `fetch <pattern>` reads 1 GiB exactly once.")
print("FETCH_SIZE is reported in KiB per dispatch (rocprofv3 counter_collection.csv, summed over its rows of a dispatch); factor = bytes read / FETCH_SIZE bytes.")
print()
for p in ("wide", "seg16k", "seg32k"):
    for kind in ("fetch", "rdreq", "tcc"):
        files = glob.glob(os.path.join(O, "%s_%s" % (kind, p), "**", "*counter_collection.csv"), recursive=True)
        if not files:
            print("%-7s %-6s no counter file (see %s_%s.log)" % (p, kind, kind, p)); continue
        per = {}
        for row in csv.DictReader(open(files[0])):
            name = row.get("Kernel_Name", "")
            if "wide_kernel" not in name and "seg_kernel" not in name:
                continue
            per.setdefault((row["Dispatch_Id"], row["Counter_Name"]), 0.0)
            per[(row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
        by_counter = {}
        for (d, c), v in per.items():
            by_counter.setdefault(c, []).append(v)
        for c, vals in sorted(by_counter.items()):
            mean = sum(vals) / len(vals)
            extra = ""
            if c == "FETCH_SIZE":
                extra = "  = %.1f MiB as KiB -> factor %.3f (bytes read / counter bytes)" % (mean / 1024.0, BYTES / (mean * 1024.0))
            elif "RDREQ" in c:
                extra = "  -> %.1f bytes read per request" % (BYTES / mean if mean else 0)
            print("%-7s %-28s dispatches %d  mean %.1f  (min %.1f max %.1f)%s" % (p, c, len(vals), mean, min(vals), max(vals), extra))
    print()
PY
cat $O/issue_slots.txt; cat $O/fetch_factor.txt
