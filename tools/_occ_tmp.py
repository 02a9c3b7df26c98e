import csv, glob, sys
from collections import defaultdict
d = sys.argv[1]
per = defaultdict(float); names = {}
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "layered_qc" not in r["Kernel_Name"] and "flood_qc" not in r["Kernel_Name"]: continue
        per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"]); names[r["Dispatch_Id"]] = r["Kernel_Name"][:60]
for disp in sorted(names, key=int)[-2:]:
    w = per[(disp, "SQ_WAVES")]; wc = per[(disp, "SQ_WAVE_CYCLES")]; gui = per[(disp, "GRBM_GUI_ACTIVE")] / 8
    print(names[disp], "waves", int(w), "ms %.1f" % (gui / 2.4e6), "occupancy %.1f waves/CU" % (wc * 4 / gui / 256), "lifetime/wave %.2f ms" % (wc * 4 / w / 2.4e6))
