import json,sys
for p in sys.argv[1:]:
    try:
        j=json.loads(open(p).read().strip().splitlines()[-1])
    except Exception as e:
        print(p,"ERR",e); continue
    r=j.get("roofline",{})
    print(p, j["config"].get("workload","")[:40], j["dtype"], "value %.3fM"%(j["value"]/1e6), "ms/step %.3f"%j["ms_per_step"], "kern", r.get("kernel"), "kern_ms", r.get("kernel_ms"), "frac %.3f"%r.get("frac",0), "traffic", r.get("traffic"))
