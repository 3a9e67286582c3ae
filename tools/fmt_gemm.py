import sys, json
print(" ".join(f"{d['name']}={d['TFLOPs']}" for d in (json.loads(l) for l in sys.stdin if l.startswith("{"))))
