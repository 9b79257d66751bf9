"""Stamp a traffic JSON written by profiles/summarize_rocpd.py --json with the identity of what was profiled: sha256 of the
libbkhip.so in the tree and the hash of the kick+drift kernel's source text (bench.py reports `roofline.traffic` only while
one of them still matches).  usage: stamp_traffic.py FILE "profiled in ..." """
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
path, where = sys.argv[1], sys.argv[2]
t = json.load(open(path))
t["lib_sha256_16"] = bench.lib_hash()
t["source_sha256_16"] = bench.source_hash()
t["profiled_in"] = where
json.dump(t, open(path, "w"), indent=1)
print("stamped", path, t["lib_sha256_16"], t["source_sha256_16"])
