import sys, os
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import flux_amd as flux
from oracle import oracle as oracle_mod
import test_gpu_fuzz as tf
from conftest import max_abs_diff
demo1 = flux.load_scene(os.path.join(root, "scenes/demo1.yml"))
bad = 0
for chunk in range(100, 100 + int(sys.argv[1])):
    try:
        tf.test_random_scenes_against_the_oracle.__wrapped__ if False else None
        tf.test_random_scenes_against_the_oracle(flux, oracle_mod, demo1, chunk)
    except AssertionError as e:
        bad += 1
        print("FAIL analytic chunk", chunk, str(e)[:300], flush=True)
    if chunk % 4 == 0:
        try:
            tf.test_random_meshes_against_the_oracle(flux, oracle_mod, demo1, chunk)
        except AssertionError as e:
            bad += 1
            print("FAIL mesh chunk", chunk, str(e)[:300], flush=True)
    if chunk % 10 == 0: print("chunk", chunk, "failures so far", bad, flush=True)
print("done, failures:", bad)
