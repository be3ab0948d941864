"""CPU: everything on the host side of the C-ABI -- no compute through the HIP library.
Library/ABI surface, CLI, schedules, checkpoint retention, loaders, JSON writer, and the
index-generation flow of the ORACLE pipeline against the reference's recorded output."""
import hashlib
import json
import os
import re
import subprocess

import numpy as np
import pytest
import torch

import golden_inputs as gi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def f7():
    with open(os.path.join(GOLD, "f7_trainer_host_logic.json")) as fh:
        return json.load(fh)


# ------------------------------------------------------------------ the C-ABI library
def test_library_exports_every_declared_symbol():
    """include/lcrec.h <-> liblcrec_hip.so <-> the ctypes table: same set of entry points."""
    import lcrec_amd
    header = open(os.path.join(ROOT, "include", "lcrec.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(lcrec_[a-z_0-9]+)\s*\(", header))
    assert declared == set(lcrec_amd._lib.EXPORTS)
    out = subprocess.check_output(["nm", "-D", "--defined-only", lcrec_amd._lib.LIB_PATH], text=True)
    exported = set(re.findall(r" T (lcrec_[a-z_0-9]+)", out))
    assert declared <= exported
    lib = lcrec_amd._lib.load()                      # loads without a GPU; nothing is launched
    assert lib.lcrec_version() == 3 == lcrec_amd._lib.ABI_VERSION
    assert lib.lcrec_last_error() == b""


def test_ops_refuse_cpu_tensors():
    import lcrec_amd
    with pytest.raises(lcrec_amd.LcrecError):
        lcrec_amd.ops.linear_forward(torch.zeros(4, 8), torch.zeros(16, 8))
    with pytest.raises(lcrec_amd.LcrecError):
        lcrec_amd.ops.rq_assign(torch.zeros(4, 32), torch.zeros(256 * 32), [256])


def test_library_argument_errors_are_reported_not_crashed():
    """Error paths that return before any launch can be exercised without a device."""
    import ctypes
    import lcrec_amd
    lib = lcrec_amd._lib.load()
    K = (ctypes.c_int * 1)(256)
    rc = lib.lcrec_rq_assign(None, 10, 32, None, K, 1, None, 0, None, 0, None, None, None, None, 0.0, None, 0, None, None)
    assert rc == -1 and b"NULL" in lib.lcrec_last_error()
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    rc = lib.lcrec_rq_assign(p, 1, 24, p, K, 1, p, 0, None, 0, None, None, None, None, 0.0, None, 0, None, None)
    assert rc == -2 and b"e_dim=24" in lib.lcrec_last_error()
    rc = lib.lcrec_rq_assign(p, 1, 32, p, K, 1, p, 0, None, 0, None, None, None, p, -1.0, None, 0, None, None)
    assert rc == -1 and b"tie_tau" in lib.lcrec_last_error()
    # the context API's argument checks (no device needed: nothing is created before first use)
    assert lib.lcrec_context_create(None) == -1
    assert lib.lcrec_context_set_pipelines(None, 2) == -1
    assert lib.lcrec_context_destroy(None) == 0
    rc = lib.lcrec_linear_forward(p, 2, 12, p, None, None, None, 0, 16, p, None)
    assert rc == -2 and b"multiple of 8" in lib.lcrec_last_error()
    # backward products: shape rules, workspace contract, and the split count is a pure function of the sizes
    assert lib.lcrec_linear_backward(None, p, p, 4, 8, 32, p, p, None, 0, None) == -1
    assert lib.lcrec_linear_backward(p, p, p, 4, 8, 48, p, None, None, 0, None) == -2 and b"multiple of 32" in lib.lcrec_last_error()
    assert lib.lcrec_linear_backward(p, p, p, 4, 6, 32, None, p, None, 0, None) == -2 and b"multiples of 4" in lib.lcrec_last_error()
    assert lib.lcrec_linear_backward_splits(2048, 768, 2048) == 1          # wide layer: enough tiles already
    s_narrow = lib.lcrec_linear_backward_splits(2048, 64, 32)
    assert 2 <= s_narrow <= 16 and lib.lcrec_linear_backward_workspace(2048, 64, 32) == s_narrow * 64 * 32 * 4
    assert lib.lcrec_linear_backward_splits(64, 64, 32) == 1               # too few K-tiles to split
    assert lib.lcrec_linear_backward(p, p, p, 2048, 64, 32, None, p, None, 0, None) == -3 and b"workspace" in lib.lcrec_last_error()
    assert lib.lcrec_linear_backward(p, p, p, 0, 64, 32, p, p, None, 0, None) == 0        # empty batch


# ------------------------------------------------------------------ module tree / state dict
@pytest.mark.parametrize("bn", [False, True])
def test_state_dict_layout_matches_reference(bn):
    import lcrec_amd
    g = np.load(os.path.join(GOLD, f"f4_step_bn{int(bn)}.npz"))
    ref = {k[4:]: g[k] for k in g.files if k.startswith("sd__")}
    model = lcrec_amd.RQVAE(in_dim=128, num_emb_list=[256] * 4, e_dim=16, layers=[64, 32], bn=bn, kmeans_init=False,
                            sk_epsilons=[0.0, 0.0, 0.0, 0.003])
    sd = model.state_dict()
    assert list(sd) == list(ref)                                       # same names in the same order
    for k, v in sd.items():
        assert tuple(v.shape) == ref[k].shape and str(v.dtype).replace("torch.", "") == str(ref[k].dtype), k
    # improve fork: EMA buffers appear under the reference's names
    ema = lcrec_amd.RQVAE(in_dim=128, num_emb_list=[32, 32], e_dim=16, layers=[64], sk_epsilons=[0.0, 0.0], ema_decay=0.99)
    assert "rq.vq_layers.0._ema_cluster_size" in ema.state_dict() and "rq.vq_layers.1._ema_w" in ema.state_dict()
    # rq.py:30: the shorter of (num_emb_list, sk_epsilons) sets the depth
    short = lcrec_amd.RQVAE(in_dim=128, num_emb_list=[256] * 4, e_dim=16, layers=[64], sk_epsilons=[0.0, 0.0, 0.0])
    assert len(short.rq.vq_layers) == 3
    # kmeans_init zeroes the codebooks and defers initialisation (vq.py:21-27)
    lazy = lcrec_amd.VectorQuantizer(64, 16, kmeans_init=True)
    assert not lazy.initted and float(lazy.embedding.weight.abs().sum()) == 0.0


# ------------------------------------------------------------------ CLI
def test_cli_defaults_and_bool_quirk_match_reference():
    from lcrec_amd import main as cli
    ref = f7()
    mine = vars(cli.parse_args([]))
    for k, v in ref["cli_defaults"].items():
        assert mine[k] == v, k
    q = cli.parse_args(["--bn", "False", "--kmeans_init", "False", "--sk_epsilons", "0.0", "0.003"])
    assert q.bn is ref["cli_bool_quirk"]["bn"] is True                 # `--bn False` is True (type=bool)
    assert q.kmeans_init is ref["cli_bool_quirk"]["kmeans_init"] is True
    assert q.sk_epsilons == ref["cli_bool_quirk"]["sk_epsilons"]
    off = cli.parse_args(["--bn", "False", "--no_bn", "--no_kmeans_init"])
    assert off.bn is False and off.kmeans_init is False
    assert mine["ema_decay"] is None                                   # index/ behaviour unless asked


# ------------------------------------------------------------------ schedules / retention
def test_lr_schedules_match_transformers_values():
    from lcrec_amd.trainer import constant_schedule_with_warmup, linear_schedule_with_warmup
    for key, want in f7()["lr_multipliers"].items():
        warm, total = (int(v) for v in key.split(","))
        p = torch.nn.Parameter(torch.zeros(1))
        o1, o2 = torch.optim.SGD([p], lr=1.0), torch.optim.SGD([p], lr=1.0)
        s1 = linear_schedule_with_warmup(o1, warm, total)
        s2 = constant_schedule_with_warmup(o2, warm)
        lin, con = [s1.get_last_lr()[0]], [s2.get_last_lr()[0]]
        for _ in range(total + 3):
            o1.step(); s1.step(); o2.step(); s2.step()
            lin.append(s1.get_last_lr()[0]); con.append(s2.get_last_lr()[0])
        assert lin == want["linear"] and con == want["constant"], key


def test_checkpoint_retention_walk_matches_reference(tmp_path):
    """Drive the Trainer's fit() with the scripted losses / collision rates the reference Trainer was driven
    with (make_golden.fixture_trainer) and compare the directory listing after every evaluation."""
    import argparse
    import logging
    from lcrec_amd import trainer as tr_mod
    import lcrec_amd
    ref = f7()["retention"]
    rates, losses = ref["collision_rates"], ref["train_losses"]
    args = argparse.Namespace(lr=1e-3, learner="AdamW", lr_scheduler_type="linear", weight_decay=1e-4, epochs=len(rates),
                              warmup_epochs=1, save_limit=ref["save_limit"], eval_step=1, device="cpu",
                              ckpt_dir=str(tmp_path))
    model = lcrec_amd.RQVAE(in_dim=32, num_emb_list=[8, 8], e_dim=16, layers=[24], bn=True, kmeans_init=False,
                            sk_epsilons=[0.0, 0.0])
    walk, snaps = [], []

    class Scripted(tr_mod.Trainer):
        def _train_epoch(self, data, epoch_idx):
            return losses[epoch_idx], losses[epoch_idx] / 2

        def _valid_epoch(self, data):
            if walk:
                snaps.append(sorted(os.listdir(self.ckpt_dir)))
            walk.append(rates[len(walk)])
            return walk[-1]

    t = Scripted(args, model, data_num=4)
    logging.disable(logging.CRITICAL)
    try:
        best = t.fit(None)
    finally:
        logging.disable(logging.NOTSET)
    snaps.append(sorted(os.listdir(t.ckpt_dir)))
    assert snaps == ref["files_after_each_eval"]
    assert list(best) == ref["fit_returns"]
    # checkpoint schema (trainer.py:158-166)
    schema = f7()["checkpoint_schema"]
    ck = torch.load(os.path.join(t.ckpt_dir, "best_collision_model.pth"), weights_only=False)
    assert sorted(ck) == schema["keys"] and type(ck["args"]).__name__ == schema["args_type"]
    assert {k: [list(v.shape), str(v.dtype)] for k, v in ck["state_dict"].items()} == schema["state_dict"]
    assert sorted(ck["optimizer"]) == schema["optimizer_keys"]
    assert ck["epoch"] == schema["epoch"] and ck["best_collision_rate"] == schema["best_collision_rate"]
    # the index generator reads the same file with the restricted unpickler (no weights_only=False fallback) ...
    from lcrec_amd import generate_indices as gen
    safe = gen.load_checkpoint(os.path.join(t.ckpt_dir, "best_collision_model.pth"))
    assert sorted(safe) == schema["keys"] and vars(safe["args"]) == vars(ck["args"])
    assert all(torch.equal(v, safe["state_dict"][k]) for k, v in ck["state_dict"].items())
    # ... and refuses a file that names any other global instead of executing it
    import pickle

    class Evil:
        def __reduce__(self):
            return (os.system, ("echo pwned > %s" % os.path.join(t.ckpt_dir, "pwned"),))
    bad = os.path.join(t.ckpt_dir, "crafted.pth")
    torch.save({"args": Evil(), "state_dict": {}}, bad, pickle_protocol=4)
    with pytest.raises(pickle.UnpicklingError):
        gen.load_checkpoint(bad)
    assert not os.path.exists(os.path.join(t.ckpt_dir, "pwned"))


# ------------------------------------------------------------------ data
def test_dataset_and_device_loader_semantics(tmp_path):
    from lcrec_amd.datasets import DeviceLoader, EmbDataset
    x = np.random.RandomState(0).standard_normal((103, 8)).astype(np.float64)     # float64 on disk: cast to fp32
    path = str(tmp_path / "x.npy")
    np.save(path, x)
    ds = EmbDataset(path)
    assert ds.dim == 8 and len(ds) == 103
    assert ds[5].dtype == torch.float32 and torch.equal(ds[5], torch.from_numpy(x[5]).float())
    assert tuple(ds[[1, 7, 3]].shape) == (3, 8)                                    # generate_indices.py:117
    # shuffled batches are the ones torch's DataLoader yields under the same global seed
    torch.manual_seed(2024)
    ref_batches = [b.clone() for b in torch.utils.data.DataLoader(ds, batch_size=16, shuffle=True, num_workers=0)]
    ref_batches += [b.clone() for b in torch.utils.data.DataLoader(ds, batch_size=16, shuffle=True, num_workers=0)]
    torch.manual_seed(2024)
    loader = DeviceLoader(ds, batch_size=16, shuffle=True, device="cpu")
    mine = [b.clone() for b in loader] + [b.clone() for b in loader]
    assert len(loader) == 7 and len(mine) == len(ref_batches)
    # (DataLoader with num_workers=0 skips the base-seed draw the worker path makes; compare as sets of rows per epoch)
    for e in range(2):
        a = torch.cat(mine[7 * e:7 * e + 7])
        assert sorted(map(tuple, a.tolist())) == sorted(map(tuple, torch.from_numpy(x).float().tolist()))
    seq = [b for b in DeviceLoader(ds, batch_size=64, shuffle=False, device="cpu")]
    assert torch.equal(torch.cat(seq), torch.from_numpy(x).float())


def test_device_loader_reproduces_worker_dataloader_order(tmp_path):
    """index/main.py:86 uses num_workers=4: the iterator draws a base seed before the sampler seeds itself."""
    from lcrec_amd.datasets import DeviceLoader, EmbDataset
    x = np.arange(40 * 4, dtype=np.float32).reshape(40, 4)
    path = str(tmp_path / "x.npy")
    np.save(path, x)
    ds = EmbDataset(path)
    torch.manual_seed(7)
    want = torch.cat([b for b in torch.utils.data.DataLoader(ds, batch_size=8, shuffle=True, num_workers=2)])
    torch.manual_seed(7)
    got = torch.cat([b for b in DeviceLoader(ds, batch_size=8, shuffle=True, device="cpu")])
    assert torch.equal(got, want)


# ------------------------------------------------------------------ index generation (oracle pipeline) + writer
def test_oracle_generate_pipeline_reproduces_reference_json(tmp_path):
    from oracle import generate_ref
    from lcrec_amd import generate_indices as gen
    g = np.load(os.path.join(GOLD, "f6_generate.npz"))
    meta = json.load(open(os.path.join(GOLD, "manifest.json")))["fixtures"]["f6_generate.npz"]
    text = bytes(g["json_text"]).decode()
    assert hashlib.sha256(text.encode()).hexdigest() == meta["json_sha256"]
    x = gi.toy_items(meta["seed"])
    sd = {k[4:]: g[k] for k in g.files if k.startswith("sd__")}
    names = gi.state_dict_names(3, False, 3)
    idx, history, out = generate_ref.run(x, [sd[n + ".weight"] for n in names["encoder"]],
                                         [sd[n + ".bias"] for n in names["encoder"]], [sd[n] for n in names["codebooks"]])
    assert history == g["groups_per_round"].tolist() and len(history) == 20       # hits the 20-round cap
    assert np.array_equal(idx, g["idx"].astype(np.int64))
    assert out == text                                                             # byte-identical .index.json
    # the product's writer emits the same bytes, and its helpers have the reference's semantics
    path = str(tmp_path / "a.index.json")
    gen.dump_index_json(idx.tolist(), path)
    assert open(path).read() == text
    keys = [tuple(r) for r in idx.tolist()]
    groups = gen.get_collision_item(keys)
    assert groups == generate_ref.collision_groups(keys)
    assert all(g_ == sorted(g_) for g_ in groups) and [g_[0] for g_ in groups] == sorted(g_[0] for g_ in groups)
    assert gen.check_collision(keys) is False and max(gen.get_indices_count(keys).values()) >= 3


def test_index_json_is_what_the_downstream_reader_expects(tmp_path):
    """data.py:38-89 contract: json.load, str item keys in order, "".join(tokens), tokens <letter_int>."""
    from lcrec_amd import generate_indices as gen
    rows = [[1, 22, 255, 0], [7, 0, 3, 19], [255, 255, 255, 255]]
    path = str(tmp_path / "t.index.json")
    gen.dump_index_json(rows, path)
    text = open(path).read()
    assert text == json.dumps({i: t for i, t in enumerate(gen.tokens_for(rows))})
    index = json.load(open(path))
    assert list(index) == ["0", "1", "2"] and "".join(index["0"]) == "<a_1><b_22><c_255><d_0>"
    new_tokens = sorted({t for toks in index.values() for t in toks})
    assert all(re.fullmatch(r"<[a-d]_\d+>", t) for t in new_tokens)
    assert gen.tokens_for([[1] * 8])[0][5:] == ["<f_1>", "<g_1>", "<h_1>"]          # beyond the reference's 5 prefixes
    # the consumer's three views (data.py:43-79) of the reference's own F6 output and of ours are the same objects
    from oracle import generate_ref
    g = np.load(os.path.join(GOLD, "f6_generate.npz"))
    ref_index = json.loads(bytes(g["json_text"]).decode())
    path2 = str(tmp_path / "f6.index.json")
    gen.dump_index_json(g["idx"].astype(np.int64), path2)
    ours = json.load(open(path2))
    assert generate_ref.reader_views(ours) == generate_ref.reader_views(ref_index)
    new_tokens, all_items, allowed = generate_ref.reader_views(ours)
    n_unique = len({tuple(r) for r in g["idx"].tolist()})
    assert len(all_items) == n_unique and sorted(allowed) == list(range(g["idx"].shape[1]))
    assert all(tok.startswith("<%s_" % "abcde"[i]) for i, toks in allowed.items() for tok in toks)


def test_native_index_json_text_equals_json_dump():
    """lcrec_index_json_format (host-side text, no device work) against json.dumps of the reference's
    dict-of-token-lists (generate_indices.py:83-92,138-145): single-thread and sliced paths, L up to 26,
    wide and negative values, a non-zero first item, and the tight-buffer error."""
    import ctypes
    from lcrec_amd import _lib, ops, generate_indices as gen
    rs = np.random.RandomState(0)
    for n, L, hi in [(1, 1, 5), (3, 4, 256), (70000, 4, 256), (66000, 8, 1024), (5, 26, 2 ** 62), (0, 4, 256)]:
        a = rs.randint(0, hi, size=(n, L), dtype=np.int64)
        if n > 3:
            a[2, 0] = -1
        want = json.dumps({7 + i: gen.tokens_for([r])[0] for i, r in enumerate(a.tolist())})[1:-1].encode()
        assert ops.index_json_text(a, first_item=7) == want, (n, L)
    lib = _lib.load()
    a = np.arange(40, dtype=np.int64).reshape(10, 4)
    buf = ctypes.create_string_buffer(64)
    assert lib.lcrec_index_json_format(a.ctypes.data, 10, 4, 0, ctypes.addressof(buf), 64) == -3
    assert b"too small" in lib.lcrec_last_error()
    assert lib.lcrec_index_json_format(a.ctypes.data, 10, 27, 0, ctypes.addressof(buf), 64) == -1
    # tight (but sufficient) buffer: the checked single-pass path gives the same bytes
    want = json.dumps({i: gen.tokens_for([r])[0] for i, r in enumerate(a.tolist())})[1:-1].encode()
    cap = len(want) + 140
    buf = ctypes.create_string_buffer(cap)
    got = lib.lcrec_index_json_format(a.ctypes.data, 10, 4, 0, ctypes.addressof(buf), cap)
    assert buf.raw[:got] == want


# ------------------------------------------------------------------ k-means init (host path)
def test_kmeans_host_path_reproduces_reference_fixture():
    """F10: lcrec_amd.layers.kmeans (sklearn on the host, the reference's own call, layers.py:69-82) under
    np.random.seed(2024) gives the centres the imported reference gave -- for the scikit-learn version the fixture
    was generated with (the reference pins none; another version is 'parity unpinned' and skipped)."""
    import sklearn
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", "manifest.json")))["fixtures"]["f10_kmeans.npz"]
    if sklearn.__version__ != meta["sklearn"]:
        pytest.skip(f"fixture generated with scikit-learn {meta['sklearn']}, this is {sklearn.__version__}")
    from lcrec_amd import layers
    g = np.load(os.path.join(ROOT, "tests", "golden", "f10_kmeans.npz"))
    x = torch.from_numpy(gi.kmeans_case())
    assert layers.KMEANS_IMPL == "sklearn"
    for K, iters in ((256, 10), (64, 100)):
        np.random.seed(2024)
        c = layers.kmeans(x, K, iters)
        assert c.dtype == torch.float32 and tuple(c.shape) == (K, 32)
        # sklearn's Lloyd reduces per-thread partial sums in arrival order: identical up to fp32 rounding across runs
        np.testing.assert_allclose(c.numpy(), g[f"centres_{K}_{iters}"], rtol=1e-4, atol=1e-5)


def test_recheck_neartie_reproduces_the_references_batch64_tuples_on_the_recorded_rows():
    """generate --recheck_neartie (hard part 1's slow path): the product's own torch-CPU restatement of
    the reference's op order, run on the 64-row batches index/generate_indices.py:77-79 forms, gives the REFERENCE's
    batch-64 tuple on every one of the 33 rows of the 1 M x 768-d audit where the canonical order differs (fixture F9, whose
    `ref64_rows` came from the imported reference in this container).  Like the fixture, the outcome belongs to this
    host's BLAS: the test skips itself where the two rows the reference itself computes differently at batch 4096 vs 64
    do not reproduce (another CPU / MKL build) -- "parity unpinned across hosts", DESIGN.md section 2.1."""
    import argparse
    import hashlib
    import lcrec_amd
    from lcrec_amd import generate_indices as gen
    f = np.load(os.path.join(GOLD, "f9_neartie_c3.npz"))
    n, in_dim = gi.NEARTIE_CASES["c3"]
    x = gi.neartie_items(n, in_dim)
    if hashlib.sha256(x[:65536].tobytes()).hexdigest() != str(f["sha_x_head"]):
        pytest.skip("numpy's PCG64 float32 normal stream differs from the fixture's")
    dims, Ws, bs = gi.neartie_encoder(in_dim)
    names = gi.state_dict_names(len(Ws), False, 4)
    sd = {}
    for l, (W, b) in enumerate(zip(Ws, bs)):
        sd[names["encoder"][l] + ".weight"], sd[names["encoder"][l] + ".bias"] = torch.from_numpy(W), torch.from_numpy(b)
    for l, c in enumerate(f["codebooks"]):
        sd[names["codebooks"][l]] = torch.from_numpy(c.copy())
    rows = f["rows"]
    args = argparse.Namespace(layers=gi.RUN_SH_LAYERS, bn=False)
    idx = torch.zeros((n, 4), dtype=torch.int64)
    idx[rows] = torch.from_numpy(f["oracle_rows"].astype(np.int64))
    flags = torch.zeros(n, dtype=torch.int32)
    flags[rows] = 1
    resid_last = torch.zeros((n, 32))
    torch.set_num_threads(8)                                    # the thread count of the fixture's run (manifest: cpu_threads)
    done, changed = gen.recheck_neartie(sd, args, x, idx, resid_last, flags)
    got, want = idx[rows].numpy(), f["ref64_rows"].astype(np.int64)
    if not np.array_equal(got, want):
        pytest.skip(f"{int((got != want).any(1).sum())} of 33 recorded rows come out differently on this host's BLAS "
                    "(the fixture's tuples are this build container's)")
    assert done == 33 and changed == int((f["ref64_rows"] != f["oracle_rows"]).any(1).sum())
    early = (f["ref64_rows"][:, :3] != f["oracle_rows"][:, :3]).any(1)
    assert bool((resid_last[rows[early]].abs().sum(1) > 0).all()) and float(resid_last[rows[~early]].abs().sum()) == 0.0
