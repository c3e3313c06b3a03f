import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=j["secondary"]
print(sys.argv[1], "value %.0f | " % j["value"] + " ".join("%s %.3g" % (k.split("_")[0][:8], v["value"]) for k, v in s.items() if isinstance(v, dict) and "value" in v))
