"""Development aid: bench.kernel_source_hash() changed its definition in round 3 (comments and line breaks no longer count).
Rewrites `kernel_source_sha` in profiles/<round>_*_traffic.json to the new definition -- only where the old-style hash of
mgm_amd/csrc at the given commit IS the sha the file carries (i.e. the file was measured on that commit's kernels).
python tools/rehash_traffic.py r03 18a7d47"""
import glob, hashlib, json, os, shutil, subprocess, sys, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench

rnd, commit = sys.argv[1], sys.argv[2]
t = tempfile.mkdtemp()
subprocess.run("git archive %s mgm_amd/csrc | tar -x -C %s" % (commit, t), shell=True, check=True, cwd=bench.ROOT)
d = os.path.join(t, "mgm_amd", "csrc")
h = hashlib.sha256()
for f in sorted(os.listdir(d)):
    if f.endswith((".hip", ".h")):
        h.update(open(os.path.join(d, f), "rb").read())
old_style, new_style = h.hexdigest()[:16], bench.kernel_source_hash(d)
shutil.rmtree(t)
for p in sorted(glob.glob(os.path.join(bench.ROOT, "profiles", rnd + "_*_traffic.json"))):
    j = json.load(open(p))
    if j.get("kernel_source_sha") == old_style:
        j["kernel_source_sha"] = new_style
        json.dump(j, open(p, "w"), indent=1)
        print(os.path.basename(p), old_style, "->", new_style)
    else:
        print(os.path.basename(p), "carries", j.get("kernel_source_sha"), "- left alone")
print("the tree now hashes to", bench.kernel_source_hash())
