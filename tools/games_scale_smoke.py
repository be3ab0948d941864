#!/usr/bin/env python3
"""End-to-end at the reference's Games size on synthetic embeddings: train (reference CLI flags of index/run.sh,
few epochs), then generate `.index.json`, then load it the way data.py does.  Prints wall times.

    python tools/games_scale_smoke.py [--epochs 30] [--items 16859] [--in_dim 4096] [--kmeans_impl sklearn|device]
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lcrec_amd import generate_indices as gen  # noqa: E402
from lcrec_amd import main as cli  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=30)
    ap.add_argument("--items", type=int, default=16859)
    ap.add_argument("--in_dim", type=int, default=4096)
    ap.add_argument("--kmeans_impl", default="sklearn")
    ap.add_argument("--train_engine", default="auto", choices=["auto", "off"])
    a = ap.parse_args()
    with tempfile.TemporaryDirectory() as tmp:
        rs = np.random.RandomState(0)
        centres = rs.standard_normal((200, a.in_dim)).astype(np.float32)
        x = centres[rs.randint(0, 200, size=a.items)] + 0.5 * rs.standard_normal((a.items, a.in_dim)).astype(np.float32)
        path = os.path.join(tmp, "Games.emb-synth-td.npy")
        np.save(path, x)
        t0 = time.perf_counter()
        best_loss, best_rate = cli.main(["--data_path", path, "--ckpt_dir", os.path.join(tmp, "ckpt"), "--device", "cuda:0",
                                         "--epochs", str(a.epochs), "--eval_step", str(max(1, a.epochs // 3)),
                                         "--batch_size", "1024", "--lr", "1e-3", "--weight_decay", "1e-4",
                                         "--num_emb_list", "256", "256", "256", "256", "--sk_epsilons", "0.0", "0.0", "0.0", "0.003",
                                         "--layers", "2048", "1024", "512", "256", "128", "64", "--e_dim", "32",
                                         "--kmeans_impl", a.kmeans_impl, "--bn", "False", "--train_engine", a.train_engine])    # run.sh:9: parses as True
        t1 = time.perf_counter()
        run = sorted(os.listdir(os.path.join(tmp, "ckpt")))[-1]
        ckpt = os.path.join(tmp, "ckpt", run, "best_collision_model.pth")
        out = os.path.join(tmp, "Games.index.json")
        stats = gen.generate(ckpt, out, device="cuda:0", data_path=path, verbose=False)
        t2 = time.perf_counter()
        index = json.load(open(out))
        steps = a.epochs * -(-a.items // 1024)
        print(f"train {a.epochs} epochs ({steps} steps; index/run.sh flags incl. `--bn False` = BatchNorm ON) incl. k-means init, 3 evals, checkpoints: {t1 - t0:.2f} s "
              f"({(t1 - t0) / steps * 1e3:.2f} ms/step all-in); best loss {best_loss:.5f}, best collision rate {best_rate:.5f}")
        print(f"generate: {t2 - t1:.2f} s, rounds {stats['rounds']}, collision rate {stats['collision_rate']:.6f}, "
              f"max conflicts {stats['max_conflicts']}; json entries {len(index)}, first {index['0']}")


if __name__ == "__main__":
    main()
