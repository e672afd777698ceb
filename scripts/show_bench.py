import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "kernel_ms", d["roofline"]["kernel_ms"], "traffic", d["roofline"].get("traffic"))
print("steady", d.get("steady_state"))
cb=d["cpu_baseline"]; print({k:cb.get(k) for k in ("value","cores","threads_used","cores_visible","logical_cpus","cpu_model")})
for k,v in d.get("other_configs",{}).items():
    print(k, {kk: v.get(kk) for kk in ("value","ms_per_step","frac","launch","error","value_is")})
    print("   steady:", v.get("steady_state"))
    print("   cpu:", {kk: (v.get("cpu_baseline") or {}).get(kk) for kk in ("value","threads_used","cores_visible","cpu_model")})
print(d.get("protocol",{}).get("order"))
