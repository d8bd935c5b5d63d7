"""Merge a freshly measured tile table into the shipped one: measured choices replace the old ones except where the old entry
was chosen inside the training step (4th element, tools/tune_in_step.py); old geometries the new run did not see are kept.
    python tools/merge_tune.py new.json shipped.json out.json"""
import json, sys
new, old = json.load(open(sys.argv[1])), json.load(open(sys.argv[2]))
e = dict(old["entries"])
kept = 0
for k, v in new["entries"].items():
    if len(e.get(k, ())) > 3 and e[k][3] and "--drop-in-step" not in sys.argv:
        kept += 1
        continue
    e[k] = v[:3]
new["entries"] = e
json.dump(new, open(sys.argv[3], "w"), indent=0, sort_keys=True)
print("entries %d, in-step entries kept %d" % (len(e), kept))
