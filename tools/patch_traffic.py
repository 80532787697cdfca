#!/usr/bin/env python3
"""tools/patch_traffic.py <cfg>...: a bench line taken before its PMC passes cannot know their
result; fill profiles/r03_bench_<cfg>.json's roofline.traffic from profiles/traffic_<cfg>.json of
the same session (same kernel sources: the hash is checked)."""
import json
import sys

sys.path.insert(0, ".")
import bench

h = bench.kernel_source_hash()
for c in sys.argv[1:]:
    p = f"profiles/r03_bench_{c}.json"
    d = json.load(open(p))
    t = json.load(open(f"profiles/traffic_{c}.json"))
    assert t["kernel_source_sha256"] == h, (c, "traffic file is for other kernel sources")
    assert t["kernel"].split("<")[0] in d["roofline"]["kernel"] or d["roofline"]["kernel"] in t["kernel"], (c, t["kernel"])
    d["roofline"]["traffic"] = t["traffic_bytes_per_launch"]
    d["roofline"]["traffic_source"] = (f"profiles/traffic_{c}.json ({t['source']}); filled in after the run from the PMC "
                                       "passes of the same session")
    open(p, "w").write(json.dumps(d) + "\n")
    print(c, d["roofline"]["frac"], d["roofline"]["traffic"])
