"""End to end on the MI355X: CLI training run -> checkpoint files -> index generation -> .index.json."""
import argparse
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _toy_npy(path, n=600, d=64, seed=3):
    rs = np.random.RandomState(seed)
    centres = rs.standard_normal((40, d)).astype(np.float32)
    x = centres[rs.randint(0, 40, size=n)] + 0.05 * rs.standard_normal((n, d)).astype(np.float32)
    x[10] = x[3]          # exact duplicates can never be separated (generate_indices.py:108 comment)
    x[11] = x[3]
    np.save(path, x.astype(np.float32))
    return x


def test_cli_train_then_generate(hip, tmp_path):
    from lcrec_amd import generate_indices as gen
    from lcrec_amd import main as cli
    data_path = str(tmp_path / "Toy.emb-test-td.npy")
    x = _toy_npy(data_path)
    ckpt_root = str(tmp_path / "ckpt")
    best_loss, best_rate = cli.main([
        "--data_path", data_path, "--ckpt_dir", ckpt_root, "--device", "cuda:0", "--epochs", "6", "--eval_step", "2",
        "--batch_size", "128", "--layers", "32", "--e_dim", "16", "--num_emb_list", "16", "16", "16",
        "--sk_epsilons", "0.0", "0.0", "0.003", "--sk_iters", "50", "--warmup_epochs", "1", "--lr_scheduler_type", "linear",
        "--weight_decay", "1e-4", "--save_limit", "2", "--kmeans_iters", "10"])
    assert np.isfinite(best_loss) and 0.0 <= best_rate <= 1.0
    runs = os.listdir(ckpt_root)
    assert len(runs) == 1
    files = sorted(os.listdir(os.path.join(ckpt_root, runs[0])))
    assert "best_loss_model.pth" in files and "best_collision_model.pth" in files
    assert any(f.startswith("epoch_") and f.endswith("_model.pth") for f in files)
    ckpt_path = os.path.join(ckpt_root, runs[0], "best_collision_model.pth")
    ckpt = gen.load_checkpoint(ckpt_path)
    assert sorted(ckpt) == ["args", "best_collision_rate", "best_loss", "epoch", "optimizer", "state_dict"]
    assert isinstance(ckpt["args"], argparse.Namespace)
    assert all(v.dtype == torch.float32 for v in ckpt["state_dict"].values())
    assert sorted(ckpt["state_dict"]) == sorted([
        "encoder.mlp_layers.1.weight", "encoder.mlp_layers.1.bias", "encoder.mlp_layers.4.weight",
        "encoder.mlp_layers.4.bias", "decoder.mlp_layers.1.weight", "decoder.mlp_layers.1.bias",
        "decoder.mlp_layers.4.weight", "decoder.mlp_layers.4.bias", "rq.vq_layers.0.embedding.weight",
        "rq.vq_layers.1.embedding.weight", "rq.vq_layers.2.embedding.weight"])

    out_file = str(tmp_path / "out" / "Toy.index.json")
    stats = gen.generate(ckpt_path, out_file, device="cuda:0", verbose=False)
    with open(out_file) as fh:
        text = fh.read()
    index = json.loads(text)
    assert list(index) == [str(i) for i in range(len(x))]                     # data.py iterates values positionally
    for toks in index.values():
        assert len(toks) == 3 and all(t[0] == "<" and t[1] == "abc"[i] and t[2] == "_" and t[-1] == ">"
                                      for i, t in enumerate(toks))
    assert text == json.dumps({int(k): v for k, v in index.items()})          # json.dump's default separators
    assert index["10"] == index["3"] == index["11"]                            # exact duplicates stay together
    assert stats["items"] == len(x) and stats["rounds"] <= 20

    # the batched conflict resolution equals the reference's one-forward-per-group loop on the same arithmetic
    model = gen.build_model_from_args(ckpt["args"], x.shape[1])
    model.load_state_dict(ckpt["state_dict"])
    model = model.to("cuda:0").eval()
    xd = torch.from_numpy(x).to("cuda:0")
    idx = model.get_indices(xd, use_sk=False)
    for q in model.rq.vq_layers[:-1]:
        q.sk_epsilon = 0.0
    if model.rq.vq_layers[-1].sk_epsilon == 0.0:
        model.rq.vq_layers[-1].sk_epsilon = 0.003
    history = []
    for _ in range(20):
        rows = [tuple(r) for r in idx.tolist()]
        if gen.check_collision(rows):
            break
        groups = gen.get_collision_item(rows)
        history.append(len(groups))
        for g in groups:
            idx[g] = model.get_indices(xd[g], use_sk=True)
    assert history == stats["groups_per_round"]
    assert gen.tokens_for(idx.tolist()) == list(index.values())


def test_ema_training_runs_and_reports_utilisation(hip, tmp_path):
    from lcrec_amd import main as cli
    data_path = str(tmp_path / "Toy.npy")
    _toy_npy(data_path, n=512)
    best_loss, best_rate = cli.main([
        "--data_path", data_path, "--ckpt_dir", str(tmp_path / "ck"), "--device", "cuda:0", "--epochs", "4",
        "--eval_step", "2", "--batch_size", "128", "--layers", "32", "--e_dim", "16", "--num_emb_list", "32", "32",
        "--sk_epsilons", "0.0", "0.0", "--ema_decay", "0.99", "--reset_interval", "3", "--no_kmeans_init"])
    assert np.isfinite(best_loss) and 0.0 <= best_rate <= 1.0
    # --reset_seed: the dead-code reset's draws (index_improve/models/vq.py:79-114; torch's global generators in the
    # reference) come from a dedicated device generator, so two runs are identical -- weights, EMA buffers, everything
    import glob
    from lcrec_amd import generate_indices as gen
    sds = []
    for run in ("a", "b"):
        cli.main(["--data_path", data_path, "--ckpt_dir", str(tmp_path / run), "--device", "cuda:0", "--epochs", "4",
                  "--eval_step", "2", "--batch_size", "128", "--layers", "32", "--e_dim", "16", "--num_emb_list", "32", "32",
                  "--sk_epsilons", "0.0", "0.0", "--ema_decay", "0.99", "--reset_interval", "3", "--no_kmeans_init",
                  "--reset_threshold", "0.02", "--reset_seed", "7"])
        sds.append(gen.load_checkpoint(glob.glob(str(tmp_path / run / "*" / "epoch_3_*"))[0])["state_dict"])
    assert all(torch.equal(sds[0][k], sds[1][k]) for k in sds[0])


def test_generate_reproduces_reference_index_json_bytes(hip, tmp_path):
    """tests/golden/f6_generate.npz: the reference's generate_indices.py (run unmodified but for its
    hard-coded paths) on a collision-heavy toy model -- 20 rounds, 606 -> 52 groups.  The product's
    batched device flow must emit the same bytes."""
    import argparse
    import golden_inputs as gi
    from lcrec_amd import generate_indices as gen
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g = np.load(os.path.join(gold, "f6_generate.npz"))
    meta = json.load(open(os.path.join(gold, "manifest.json")))["fixtures"]["f6_generate.npz"]
    x = gi.toy_items(meta["seed"])
    npy = str(tmp_path / "Toy.emb.npy")
    np.save(npy, x)
    kw = {k: v for k, v in meta["model"].items() if k != "in_dim"}
    args = argparse.Namespace(data_path=npy, num_workers=0, **kw)
    sd = {k[4:]: torch.from_numpy(g[k].copy()) for k in g.files if k.startswith("sd__")}
    ckpt = str(tmp_path / "toy.pth")
    torch.save({"args": args, "epoch": 0, "best_loss": 0.0, "best_collision_rate": 0.0, "state_dict": sd,
                "optimizer": {}}, ckpt, pickle_protocol=4)
    out = str(tmp_path / "Toy.index.json")
    stats = gen.generate(ckpt, out, device="cuda:0", verbose=False)
    assert stats["groups_per_round"] == g["groups_per_round"].tolist()
    want = bytes(g["json_text"])
    got = open(out, "rb").read()
    assert got == want, "index.json differs from the reference's bytes"
    # --recheck_neartie: the flagged items go through the reference's torch CPU op order on their batch-64 neighbours.  On this
    # toy every flagged item already carries the reference's tuple, so -- on a host whose BLAS rounds like the fixture's --
    # nothing changes; on another host at most the flagged items may (the slow path follows the HOST's
    # arithmetic, as a CPU run of the reference there would).
    out2 = str(tmp_path / "Toy.recheck.index.json")
    stats2 = gen.generate(ckpt, out2, device="cuda:0", verbose=False, recheck=True)
    assert stats2["rechecked_items"] == stats2["neartie_items"] == stats["neartie_items"]
    assert 0 <= stats2["recheck_changed"] <= stats2["rechecked_items"]
    a, b = json.loads(got), json.load(open(out2))
    differing = [k for k in a if a[k] != b[k]]
    if stats2["recheck_changed"] == 0:
        assert open(out2, "rb").read() == want
    else:
        assert list(b) == list(a) and len(differing) <= 20 * max(1, stats2["recheck_changed"])     # (rounds propagate a change within its groups)


@pytest.mark.parametrize("strict", [False, True])
def test_nan_loss_is_reported_with_and_without_the_per_step_sync(hip, tmp_path, strict):
    """trainer.py:40-42 raises ValueError('Training loss is nan').  The default trainer reads the flag back
    asynchronously (one step late, or when the epoch loop drains); --strict_nan_check keeps the reference's
    host sync per step.  Either way the epoch must not end silently -- also when the NaN is in the LAST batch."""
    from lcrec_amd import main as cli
    from lcrec_amd.datasets import DeviceLoader
    from lcrec_amd.trainer import Trainer
    argv = ["--data_path", "unused", "--ckpt_dir", str(tmp_path), "--device", "cuda:0", "--batch_size", "64", "--epochs", "2",
            "--layers", "32", "--e_dim", "16", "--num_emb_list", "32", "32", "--sk_epsilons", "0.0", "0.003",
            "--no_kmeans_init", "--no_bn"] + (["--strict_nan_check"] if strict else [])
    args = cli.parse_args(argv)
    cli.seed_everything(2024)
    for bad_batch in (1, 3):                       # a middle batch, and the last one
        model = cli.build_model(args, 64)
        data = torch.randn((256, 64), device="cuda:0")
        loader = DeviceLoader(data, 64, False, "cuda:0")
        trainer = Trainer(args, model, len(loader))
        trainer._train_epoch(loader, 0)            # a clean epoch passes
        data[bad_batch * 64 + 5, 7] = float("nan")
        with pytest.raises(ValueError, match="Training loss is nan"):
            trainer._train_epoch(loader, 1)


def test_bench_line_and_its_collective_path_over_a_one_rank_rccl_group(hip):
    """bench.py end to end on a small shard: the JSON contract (metric, roofline, near-tie and rank fields) and -- with
    --rehearse-rccl -- the N>1 code path (barrier, MAX all-reduce of the time, all-gather of the per-rank checksums) through
    a real RCCL communicator of one rank."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--items", "150000", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline", "--rehearse-rccl", "--dp-graph"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["ranks_seen"] == 1 and len(line["idx_checksums"]) == 1
    assert line["unit"] == "items/s" and line["value"] > 1e6 and line["scaling"] == "weak" and line["dtype"] == "f32"
    r = line["roofline"]
    assert r["bound"] == "mfma" and 0.3 < r["frac"] < 1.0 and r["kernel"] == "linear_fwd_pp_256x128"
    assert line["parity_mismatch_rows"] == 0 and 0 <= line["neartie_rows"] < 0.01 * 150000
    sec = line["secondary"]             # SURVEY.md section 8d's secondary figures ride along, outside the timed region
    assert "error" not in sec and sec["quantizer_only_items_per_s"] > 1e8 and 0.3 < sec["train_step_ms"] < 20.0
    rq = line["roofline_rq_assign"]     # the quantiser's own roofline object (north_star: MFMA utilisation on the distance GEMM)
    assert rq["kernel"] == "rq_assign" and 0.2 < rq["frac"] < 1.0 and rq["hbm_gbps_algorithmic"] > 50
    dp = line["secondary_dp"]           # the data-parallel training step through the one-rank group: eager and captured exchanges
    assert "error" not in dp and dp["train_ranks_seen"] == 1 and dp["dp_collectives_per_step"] >= 20
    assert dp["train_param_checksums_agree_eager"] and dp["train_param_checksums_agree_graph"]
    assert 0.3 < dp["dp_train_step_ms_eager"] < 50 and dp["dp_train_graph_replays_graph"] > 0 and dp["dp_train_graph_replays_eager"] == 0


@pytest.mark.parametrize("dtype,mmap", [(np.float32, True), (np.float16, True), (np.float64, False)])
def test_embedding_file_reaches_hbm_through_the_pinned_ring(tmp_path, dtype, mmap):
    """EmbDataset.to_device on a GPU (host threads fill a ring of pinned chunks, H2D copies queued in order): every row of the
    file, cast to fp32, for chunk sizes that do and do not divide the row count, more chunks than ring slots, a rank's row range."""
    from lcrec_amd.datasets import EmbDataset
    rs = np.random.RandomState(5)
    a = rs.standard_normal((1237, 96)).astype(dtype)
    path = str(tmp_path / "T.emb-x-td.npy")
    np.save(path, a)
    want = torch.from_numpy(a.astype(np.float32))
    for chunk, workers, stages in ((None, None, None), (100, 3, 4), (1, 2, 2), (1237, 8, 3), (5000, 1, 2)):
        ds = EmbDataset(path, mmap=mmap)
        got = ds.to_device("cuda:0", chunk_rows=chunk, workers=workers, stages=stages)
        assert got.dtype == torch.float32 and torch.equal(got.cpu(), want), (chunk, workers, stages)
        assert ds.to_device("cuda:0") is got                                     # kept: the loader and the index pass share it
        part = EmbDataset(path, mmap=mmap).to_device("cuda:0", chunk_rows=chunk, rows=(300, 1001), workers=workers, stages=stages)
        assert torch.equal(part.cpu(), want[300:1001])
    assert EmbDataset(path, mmap=mmap).to_device("cuda:0", rows=(7, 7)).shape == (0, 96)
