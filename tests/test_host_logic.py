"""Host-side pieces of the training step that run without a GPU: the CutMix rectangles the trainer draws must be the
oracle's maps (train.py:222-251) under the same RNG streams, draw for draw."""
import random

import numpy as np

from oracle import host_ref as H


def test_cutmix_rectangles_follow_the_reference_draw_order():
    from ustrun import trainer as T
    for seed in range(6):
        random.seed(seed); np.random.seed(seed)
        ref = [H.cutmix_box(96, p=0.6) for _ in range(12)]
        tail_ref = (random.random(), np.random.rand())
        random.seed(seed); np.random.seed(seed)
        rects = [T.cutmix_rect(96, p=0.6) for _ in range(12)]
        tail = (random.random(), np.random.rand())
        assert tail == tail_ref                                    # both RNG streams advanced identically
        for r, m in zip(rects, ref):
            assert np.array_equal(T.rect_map(r, 96), m)
        random.seed(seed); np.random.seed(seed)
        assert all(np.array_equal(T.cutmix_box(96, p=0.6), m) for m in ref)


def test_all_cover_rectangle():
    from ustrun import trainer as T
    rng = np.random.default_rng(3)
    for _ in range(20):
        region = (rng.random((40, 40)) > 0.97).astype(np.float32)
        if region.sum() == 0:
            continue
        assert np.array_equal(T.all_cover_box(region), H.all_cover_box(region))
    random.seed(1); np.random.seed(1)
    ref = H.all_cover_box(np.zeros((40, 40), dtype=np.float32))
    random.seed(1); np.random.seed(1)
    assert np.array_equal(T.all_cover_box(np.zeros((40, 40), dtype=np.float32)), ref)


def test_oracle_validation_averages_like_the_reference():
    """eval_ref.validate: batch Dice -> mean per domain loader -> mean over domains (train.py:318-372)."""
    import torch
    from oracle import eval_ref as E
    from oracle import step_ref as S
    from oracle import unet_ref as U
    from ustrun import synthetic
    torch.manual_seed(2)
    sd = U.make_state_dict(1, 2, base=4)
    loaders = synthetic.test_loaders("prostate", 2, 2, 2, 1, 32, seed=9)
    val, dom = E.validate("prostate", sd, loaders)
    manual = []
    for loader in loaders:
        ds = []
        for image, label in loader:
            with torch.no_grad():
                out = U.unet_forward(image, sd, train=False)
            pred = out.argmax(1)                              # softmax is monotone: same first-index arg-max
            ds.append(S.sample_dice("prostate", np.asarray(pred), S.decode_labels("prostate", label))[0])
        manual.append(sum(ds) / len(ds))
    assert np.allclose([d[0] for d in dom], manual) and np.isclose(val[0], sum(manual) / 2)
    lg = torch.zeros(1, 3, 2, 2); lg[0, 1] = 1; lg[0, 2] = 1
    assert torch.equal(E.predict("MNMS", lg), torch.ones(1, 2, 2, dtype=torch.long))      # first index on ties


def _load_driver(name):
    import importlib.util
    import os
    import sys
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ust-run_amd")
    for m in ("train", name):
        sys.modules.pop(m, None)
    spec = importlib.util.spec_from_file_location(name, os.path.join(root, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    sys.path.insert(0, root)
    try:
        spec.loader.exec_module(mod)
    finally:
        sys.path.remove(root)
    return mod


def test_driver_command_lines_keep_the_reference_flags_and_defaults():
    """train.py:38-79, train_mnms.py:38-77, test.py:19-32 -- flag names and defaults (new flags are additive)."""
    tr = _load_driver("train").parser.parse_args([])
    want = dict(dataset="BUSI", save_name="debug", model="unet", max_iterations=60000, num_eval_iter=500, deterministic=1,
                base_lr=0.03, seed=1337, gpu="0", threshold=0.95, amp=1, label_bs=4, unlabel_bs=4, test_bs=1, domain_num=6,
                lb_domain=1, lb_num=40, lb_ratio=0, ema_decay=0.99, consistency_type="mse", consistency=1.0,
                consistency_rampup=200.0, depth=28, widen_factor=2, leaky_slope=0.1, bn_momentum=0.1, dropout=0.0,
                cutmix_prob=1.0, LB=0.01, increase=1.0005, queue_len=10, load=False, eval=False, overwrite=False,
                load_path="../model/lb1_ratio0.2/iter_6000.pth")        # train.py:51
    for k, v in want.items():
        assert getattr(tr, k) == v, k
    mn = _load_driver("train_mnms").parser.parse_args([])
    assert (mn.dataset, mn.domain_num, mn.lb_num, mn.load_path) == ("MNMS", 4, 20, "../model/lb1_ratio0.2/iter_6000.pth")
    assert (mn.label_bs, mn.unlabel_bs, mn.queue_len, mn.threshold) == (4, 4, 10, 0.95)
    te = _load_driver("test").parser.parse_args([])
    assert (te.dataset, te.save_name, te.model, te.gpu, te.eval, te.test_bs, te.domain_num, te.lb_domain, te.save_img) == \
        ("prostate", "debug", "unet", "0", True, 1, 6, 1, False)


def test_amp_flag_selects_the_precision_like_the_reference():
    """train.py:54,551-552,842-848: `--amp 1` (the default) = fp16 autocast + GradScaler, `--amp 0` = fp32.  Here: --amp 1 ->
    the IEEE-half build with the device-side loss scale, --amp_dtype bf16 -> bfloat16, --amp 0 -> the exact path: f32 tensors
    with three-term products on the matrix cores for the U-Net (f32x3), the f32-MFMA path for DeepLabV2;
    --backend_dtype overrides (f32 stays reachable)."""
    T = _load_driver("train")
    dt = lambda *argv: T.compute_dtype(T.parser.parse_args(list(argv)))
    assert dt() == "f16" and dt("--amp", "1") == "f16" and dt("--amp", "0") == "f32x3"
    assert dt("--amp", "0", "--model", "deeplabv2") == "f32" and dt("--amp", "0", "--backend_dtype", "f32") == "f32"
    assert dt("--amp_dtype", "bf16") == "bf16" and dt("--amp", "0", "--amp_dtype", "bf16") == "f32x3"
    assert dt("--backend_dtype", "bf16") == "bf16" and dt("--amp", "0", "--backend_dtype", "f16") == "f16"


def test_load_resumes_from_the_runs_own_checkpoint_like_the_reference():
    """train.py:542-546: `--load` restores '../model/{dataset}/{save_name}/checkpoint.pth'; `--load_path` is parsed and never
    read.  The build's train() must form the same path (checked on the source: train() needs a GPU to run)."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "ust-run_amd", "train.py")).read()
    body = src[src.index("    if args.load:"):]
    body = body[:body.index("max_epoch =")]
    assert "'../model/{}/{}/checkpoint.pth'.format(args.dataset, args.save_name)" in body
    assert "args.load_path" not in body


def test_deeplab_mirror_keeps_the_reference_state_dict_surface():
    """networks/deeplabv2.py + networks/backbone/resnet.py: 626 (resnet101) state_dict entries named like the reference's
    (backbone.layer3.22.bn3.running_var ... classifier.3.bias), 42.6 M + classifier parameters, dilation plan of
    replace_stride_with_dilation=[False, True, True] (resnet.py:193-200)."""
    import torch
    from networks.deeplabv2 import DeepLabV2
    torch.manual_seed(0)
    m = DeepLabV2("resnet101", 2, pretrained=False)
    sd = m.state_dict()
    assert len(sd) == 632, len(sd)
    for k in ("backbone.conv1.weight", "backbone.layer1.0.downsample.1.running_mean", "backbone.layer3.22.bn3.running_var",
              "backbone.layer4.2.conv3.weight", "classifier.0.weight", "classifier.3.bias"):
        assert k in sd, k
    assert tuple(sd["classifier.2.weight"].shape) == (2, 2048, 3, 3)
    b = m.backbone
    assert (b.layer2[0].conv2.stride, b.layer3[0].conv2.stride, b.layer4[0].conv2.stride) == ((2, 2), (1, 1), (1, 1))
    assert (b.layer3[0].conv2.dilation, b.layer3[1].conv2.dilation, b.layer4[0].conv2.dilation, b.layer4[1].conv2.dilation) == \
        ((1, 1), (2, 2), (2, 2), (4, 4))
    assert [c.dilation[0] for c in m.classifier] == [6, 12, 18, 24] and all(c.padding == c.dilation for c in m.classifier)


def test_statistics_rows_never_exceed_the_published_bound():
    """callers size the BatchNorm-statistics buffer with ustrun_conv_mtiles; every kernel family that may serve
    the launch must stay inside it.  Host-only sweep of the row counts (ustrun_debug_conv_stat_rows builds the same launch
    description as the entry points and asks the dispatcher, without launching): the U-Net's layer shapes at every patch size
    the drivers accept, the DeepLabV2-ResNet layer shapes (1x1, dilated, strided, 64 -> 64 at the MNMS patch: the shape that
    overflowed in round 2 -- 432 rows into 400 at N = 8, 72 x 72), pooled sources, both dtypes."""
    from ustrun import _lib as L
    lib = L.lib()
    worst = 0.0
    n = 0
    for dt in (L.F32, L.BF16):
        for N in (1, 2, 3, 4, 8, 9, 16, 17, 32, 64):
            for H, W in [(h, w) for h in (8, 16, 18, 24, 33, 36, 48, 64, 65, 72, 96, 128, 129, 144, 192, 256, 288) for w in (h, h + 8, 2 * h + 2)]:
                for Cin, Cout, k, st, dil, pooled in [(64, 64, 3, 1, 1, 0), (3, 64, 3, 1, 1, 0), (1, 64, 3, 1, 1, 0), (64, 128, 3, 1, 1, 1),
                                                      (128, 64, 3, 1, 1, 0), (128, 128, 3, 1, 1, 0), (256, 256, 3, 1, 2, 0),
                                                      (512, 512, 3, 1, 4, 0), (64, 256, 1, 1, 1, 0), (256, 64, 1, 1, 1, 0),
                                                      (1024, 256, 1, 1, 1, 0), (128, 128, 3, 2, 1, 0), (256, 512, 1, 2, 1, 0),
                                                      (192, 64, 1, 1, 1, 0), (72, 24, 3, 1, 1, 0)]:
                    used = lib.ustrun_debug_conv_stat_rows(N, H, W, Cin, Cout, k, st, dil, pooled, dt)
                    cap = lib.ustrun_conv_mtiles(N, H, W, Cout)
                    assert 0 < used <= cap, (dt, N, H, W, Cin, Cout, k, st, dil, pooled, used, cap)
                    worst = max(worst, used / cap)
                    n += 1
    assert n > 5000 and worst == 1.0       # the bound is tight somewhere: it is a bound of the kernels, not a padded guess
    # round 6: the 64 -> 64 streaming kernel's flat plan (two row slots per strip: 4 N sx rows) at the image counts that select it
    for N in (33, 41, 65, 81, 129, 200):
        for H, W in ((256, 256), (288, 288), (384, 384), (512, 512), (72, 80), (200, 264), (16, 512)):
            used, cap = lib.ustrun_debug_conv_stat_rows(N, H, W, 64, 64, 3, 1, 1, 0, L.BF16), lib.ustrun_conv_mtiles(N, H, W, 64)
            assert 0 < used <= cap, (N, H, W, used, cap)
    assert lib.ustrun_debug_conv_stat_rows(81, 256, 256, 64, 64, 3, 1, 1, 0, L.BF16) == 81 * 8 * 4       # the student's call of configs[1]: flat
    # the round-2 overflow shape, by name
    assert lib.ustrun_debug_conv_stat_rows(8, 72, 72, 64, 64, 3, 1, 1, 0, L.BF16) == 432 <= lib.ustrun_conv_mtiles(8, 72, 72, 64)
    # linear tiles (round 5): one row per 256 positions of the flat padded space (19 x 19 per 18 x 18 image), one pass here
    assert lib.ustrun_debug_conv_stat_rows(64, 18, 18, 1024, 1024, 3, 1, 1, 0, L.BF16) == -(-64 * 19 * 19 // 256) <= lib.ustrun_conv_mtiles(64, 18, 18, 1024)
    # ... and not for a grid of a few blocks (one validation image): the 8 x 16 rectangular tile's rows
    assert lib.ustrun_debug_conv_stat_rows(1, 18, 18, 1024, 1024, 3, 1, 1, 0, L.BF16) == 3 * 2


def test_bench_prints_one_compact_parseable_line():
    """VERDICT r5 weak 1 / ADVICE r5: the driver parses a bounded tail of stdout, so the line bench.py prints is built by
    compact_line() from the full result and must stay small, strict JSON, and carry `roofline` + `cpu_baseline` -- checked on
    a RECORDED full result of a real run (profiles/r05c_bench.json: 21 KB with the per-layer tables)."""
    import importlib.util
    import json
    import os
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    full = json.load(open(os.path.join(ROOT, "profiles", "r05c_bench.json")))
    assert len(json.dumps(full)) > 15000 and "layers_bwd" in full["roofline"]
    line = bench.compact_line(full)
    assert "\n" not in line and len(line) < bench.LINE_LIMIT <= 6000
    got = json.loads(line, parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))     # no NaN / Infinity
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in got, k
    assert got["value"] == full["value"] and got["ms_per_step"] == full["ms_per_step"]
    r = got["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "wgrad", "doubleconv"):
        assert k in r, k
    assert "layers" not in r and "layers_bwd" not in r
    assert {b["block"] for b in r["doubleconv"]} == {"inc", "up4.conv"}
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in got["cpu_baseline"], k
    assert len(got["secondary"]) == len(full["secondary"])
    assert all("images_per_s" in s and "conv_frac" in s for s in got["secondary"])
    # a result that would still be too long sheds its optional parts instead of printing an unparseable line
    fat = dict(full, secondary=full["secondary"] * 40)
    line2 = bench.compact_line(fat)
    assert len(line2) < bench.LINE_LIMIT and "roofline" in json.loads(line2) and "cpu_baseline" in json.loads(line2)
    # a non-finite number must fail loudly here, not at the driver
    import pytest
    with pytest.raises(ValueError):
        bench.compact_line(dict(full, value=float("nan")))


def test_dsbn_mirror_keeps_the_reference_surface():
    """networks/dsbn.py:4-33: `bns` = one BatchNorm2d per domain (state_dict keys bns.<d>.*), the FIRST label selects; the UNet
    option replaces every BatchNorm2d and leaves the convolutions' initial weights (RNG order) alone."""
    import torch
    from networks.dsbn import DomainSpecificBatchNorm2d
    from networks.unet_model import UNet
    m = DomainSpecificBatchNorm2d(8, 3)
    assert sorted(m.state_dict().keys())[:2] == ["bns.0.bias", "bns.0.num_batches_tracked"] and len(m.bns) == 3
    assert m.select(torch.tensor([2, 0, 1])) is m.bns[2] and m.select([1]) is m.bns[1]
    torch.manual_seed(3); a = UNet(3, 2, base_channels=8)
    torch.manual_seed(3); b = UNet(3, 2, base_channels=8, num_domains=2)
    sa, sb = a.state_dict(), b.state_dict()
    assert len(sb) == len(sa) + 18 * 5 and all(torch.equal(sa[k], sb[k]) for k in sa if k in sb)
    assert "inc.double_conv.1.bns.1.running_var" in sb and "inc.double_conv.1.weight" not in sb
