import json,sys
for line in open(sys.argv[1]):
    line=line.strip()
    if not line.startswith("{"): 
        if line.startswith("####"): print(line)
        continue
    d=json.loads(line)
    r=d["contract_repeats"]
    print("first kernel %.1f us, repeats %s | ms_per_step first %.2f us; steady %.2f" % (r["kernel_ms_per_region"]["first_region"]*1e3, [round(x*1e3,1) for x in (r["kernel_ms_per_region"]["min"], r["kernel_ms_per_region"]["median"], r["kernel_ms_per_region"]["max"])], d["ms_per_step"]*1e3, d["steady_state"]["ms_per_step"]*1e3))
