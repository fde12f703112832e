import json,sys
d=json.loads(sys.stdin.read())
print("value %.4g ms/step %.5f refits %s" % (d["value"], d["ms_per_step"], d["forward_refits"]), {k.replace("hml_k_",""):v["kernel_us"] for k,v in d["kernels"].items()}, "two %.4g" % d.get("two_chains_one_gpu",{}).get("value",0))
