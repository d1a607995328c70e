#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel_stats + counter_collection) of one run directory into text."""
import csv, glob, sys, collections, os

def main():
    root = sys.argv[1]
    kern = sys.argv[2] if len(sys.argv) > 2 else "qr_render_kernel<false>"
    for f in sorted(glob.glob(os.path.join(root, "**", "*_kernel_stats.csv"), recursive=True)):
        print("#", os.path.relpath(f, root))
        print(open(f).read().strip())
    for f in sorted(glob.glob(os.path.join(root, "**", "*_counter_collection.csv"), recursive=True)):
        acc = collections.defaultdict(list); meta = None
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta = dict(vgpr=r["VGPR_Count"], sgpr=r["SGPR_Count"], scratch=r["Scratch_Size"],
                            lds=r["LDS_Block_Size"], grid=r["Grid_Size"], wg=r["Workgroup_Size"])
        print("#", os.path.relpath(f, root), "kernel", kern, meta)
        for k in sorted(acc):
            v = acc[k]
            print(f"{k:28s} mean {sum(v)/len(v):16.1f}  n {len(v)}")

if __name__ == "__main__":
    main()
