import json,sys
d=json.load(open(sys.argv[1]))
print(d["value"],d["ms_per_step"],{k:round(v,1) for k,v in d["phases_us_per_step"].items() if v}, round(d["roofline"]["frac"],3), round(d["roofline_materialised"]["frac"],3), round(d["roofline_materialised"]["avg_launch_us"],2))
