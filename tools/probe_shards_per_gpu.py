"""S sequence shards on one GPU (bench.py's shards_per_gpu leg on its own): python tools/probe_shards_per_gpu.py [out.json]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import bench

if __name__ == "__main__":
    r = bench.shards_per_gpu_bench(0, 0xC0FFEE)
    print(json.dumps(r, indent=1))
    if len(sys.argv) > 1:
        json.dump(r, open(sys.argv[1], "w"), indent=1)
