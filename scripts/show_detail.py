import json,sys
d=json.load(open("bench_detail.json"))["configs"]
def walk(k,v):
    if isinstance(v,dict) and ("encode" in v and "decode" in v):
        print(k)
        for dr in ("encode","decode"):
            print("  ",dr, v[dr].get("GBps"), v[dr].get("ms"), v[dr].get("kernels_ms"))
        if "single_column_latency" in v: print("   lat", v["single_column_latency"])
    elif isinstance(v,dict):
        for kk,vv in v.items(): walk(k+"."+kk,vv)
for k,v in d.items(): walk(k,v)
