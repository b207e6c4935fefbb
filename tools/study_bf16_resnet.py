#!/usr/bin/env python3
"""Study (dev tool, CPU only, no reference needed): why a random-init DeepLabV2-ResNet in TRAIN mode sits ~0.4 rel-L2 from f32 as
soon as ANY tensor on its path is rounded to bf16 -- and why keeping the residual stream in f32 does not
change that.  Emulates the rounding sites one at a time on the CPU oracle (oracle/deeplab_ref.py):

  part 1  which site matters: conv operands / raw conv outputs y1, y2 / y3 + shortcut / block outputs, alone and together;
  part 2  the proposed cure: block outputs kept in f32, the 1x1 consumers' operands centred per channel before rounding
          (train-mode BatchNorm removes the constant exactly), classifier in f32;
  part 3  per-block error growth with bf16 operands only: the error is multiplied by ~1.3 per bottleneck whatever the map size
          or batch (std/|mean| per channel printed beside it: ~1, there is no mean problem).

Log: profiles/r03_study_bf16_resnet.log.   Usage: python tools/study_bf16_resnet.py [resnet50|resnet101]"""
import sys, torch, torch.nn.functional as F
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'ust-run_amd'))
from oracle import deeplab_ref as D
from oracle.unet_ref import bn_relu
torch.set_num_threads(8)
r16 = lambda t: t.bfloat16().float()
def rel(a,b): return float((a.double()-b.double()).norm()/b.double().norm())

def fwd(x, sd, arch, train, mode):
    # mode: dict of flags: 'ops' round conv operands; 'raw' round raw conv outputs y1,y2; 'raw3' round y3/yd; 'blk' round block outputs
    def conv(inp, w, *a):
        if mode.get('ops'): inp, w = r16(inp), r16(w)
        return F.conv2d(inp, w, None, *a)
    def q(t, key): return r16(t) if mode.get(key) else t
    p0 = "backbone."
    y = q(conv(x, sd[p0+"conv1.weight"], 2, 3), 'raw')
    a = F.max_pool2d(bn_relu(y, p0+"bn1", sd, train), 3, 2, 1)
    a = q(a, 'blk')
    rate = 1
    for li,(nblk,stride,dilate) in enumerate(zip(D.ARCH[arch], (1,2,2,2), (False,False,True,True)),1):
        first_rate = rate
        if dilate: rate, stride = rate*stride, 1
        for b in range(nblk):
            p = f"{p0}layer{li}.{b}"
            s_, d_ = (stride if b==0 else 1), (first_rate if b==0 else rate)
            o = bn_relu(q(conv(a, sd[p+".conv1.weight"]), 'raw'), p+".bn1", sd, train)
            o = bn_relu(q(conv(o, sd[p+".conv2.weight"], s_, d_, d_), 'raw'), p+".bn2", sd, train)
            o = bn_relu(q(conv(o, sd[p+".conv3.weight"]), 'raw3'), p+".bn3", sd, train, relu=False)
            idn = a
            if p+".downsample.0.weight" in sd:
                idn = bn_relu(q(conv(a, sd[p+".downsample.0.weight"], s_), 'raw3'), p+".downsample.1", sd, train, relu=False)
            a = q(torch.relu(o+idn), 'blk')
    out=None
    for i,d in enumerate((6,12,18,24)):
        w = sd[f"classifier.{i}.weight"]; inp=a
        if mode.get('ops'): inp, w = r16(inp), r16(w)
        o = F.conv2d(inp, w, sd[f"classifier.{i}.bias"], 1, d, d)
        out = o if out is None else out+o
    return F.interpolate(out, size=x.shape[-2:], mode="bilinear", align_corners=True)

arch = sys.argv[1] if len(sys.argv)>1 else "resnet50"
sd = D.make_state_dict(arch, 4, 21)
g = torch.Generator().manual_seed(3)
x = torch.randn(2,3,72,104,generator=g)
with torch.no_grad():
    ref = fwd(x, {k:v.clone() for k,v in sd.items()}, arch, True, {})
    for name, mode in [("all bf16", dict(ops=1,raw=1,raw3=1,blk=1)),
                       ("blk f32", dict(ops=1,raw=1,raw3=1)),
                       ("blk+raw3 f32", dict(ops=1,raw=1)),
                       ("only ops", dict(ops=1)),
                       ("only blk", dict(blk=1)),
                       ("only raw", dict(raw=1)),
                       ("only raw3", dict(raw3=1)),
                       ("raw3+blk bf16, ops f32", dict(raw3=1, blk=1))]:
        got = fwd(x, {k:v.clone() for k,v in sd.items()}, arch, True, mode)
        print(f"{arch} train {name:28s} rel-L2 {rel(got,ref):.3e}", flush=True)

def fwd_centered(x, sd, arch, train, cls_f32=True, center=True):
    def conv(inp, w, *a):
        return r16(F.conv2d(r16(inp), r16(w), None, *a))
    def cen(a):
        return a - a.mean((0,2,3), keepdim=True) if center else a
    p0 = "backbone."
    y = conv(x, sd[p0+"conv1.weight"], 2, 3)
    a = r16(F.max_pool2d(bn_relu(y, p0+"bn1", sd, train), 3, 2, 1))
    rate = 1
    for li,(nblk,stride,dilate) in enumerate(zip(D.ARCH[arch], (1,2,2,2), (False,False,True,True)),1):
        first_rate = rate
        if dilate: rate, stride = rate*stride, 1
        for b in range(nblk):
            p = f"{p0}layer{li}.{b}"
            s_, d_ = (stride if b==0 else 1), (first_rate if b==0 else rate)
            ac = cen(a)
            o = bn_relu(conv(ac, sd[p+".conv1.weight"]), p+".bn1", sd, train)
            o = bn_relu(conv(o, sd[p+".conv2.weight"], s_, d_, d_), p+".bn2", sd, train)
            o = bn_relu(conv(o, sd[p+".conv3.weight"]), p+".bn3", sd, train, relu=False)
            idn = a
            if p+".downsample.0.weight" in sd:
                idn = bn_relu(conv(ac, sd[p+".downsample.0.weight"], s_), p+".downsample.1", sd, train, relu=False)
            a = torch.relu(o+idn)        # f32 stream
    out=None
    for i,d in enumerate((6,12,18,24)):
        w = sd[f"classifier.{i}.weight"]; inp=a
        if not cls_f32: inp, w = r16(inp), r16(w)
        o = F.conv2d(inp, w, sd[f"classifier.{i}.bias"], 1, d, d)
        out = o if out is None else out+o
    return F.interpolate(out, size=x.shape[-2:], mode="bilinear", align_corners=True)

with torch.no_grad():
    for name, kw in [("centered f32 stream, cls f32", {}), ("centered, cls bf16", dict(cls_f32=False)), ("f32 stream uncentered, cls f32", dict(center=False))]:
        got = fwd_centered(x, {k:v.clone() for k,v in sd.items()}, arch, True, **kw)
        print(f"{arch} train {name:34s} rel-L2 {rel(got,ref):.3e}", flush=True)


# ---- part 3: per-block growth of the error with bf16 operands only, at three (batch, extent) combinations ----
def relc(a, b):
    a = a - a.mean((0, 2, 3), keepdim=True); b = b - b.mean((0, 2, 3), keepdim=True)
    return rel(a, b)


def growth(N, H, W):
    gg = torch.Generator().manual_seed(3)
    xx = torch.randn(N, 3, H, W, generator=gg)

    def run(rnd):
        rec = []
        def conv(inp, w, *a):
            if rnd: inp, w = r16(inp), r16(w)
            return F.conv2d(inp, w, None, *a)
        p0 = "backbone."; sdd = {k: v.clone() for k, v in sd.items()}
        a = F.max_pool2d(bn_relu(conv(xx, sdd[p0 + "conv1.weight"], 2, 3), p0 + "bn1", sdd, True), 3, 2, 1)
        rate = 1
        for li, (nblk, stride, dilate) in enumerate(zip(D.ARCH[arch], (1, 2, 2, 2), (False, False, True, True)), 1):
            fr = rate
            if dilate: rate, stride = rate * stride, 1
            for b in range(nblk):
                p = f"{p0}layer{li}.{b}"; s_, d_ = (stride if b == 0 else 1), (fr if b == 0 else rate)
                o = bn_relu(conv(a, sdd[p + ".conv1.weight"]), p + ".bn1", sdd, True)
                o = bn_relu(conv(o, sdd[p + ".conv2.weight"], s_, d_, d_), p + ".bn2", sdd, True)
                o = bn_relu(conv(o, sdd[p + ".conv3.weight"]), p + ".bn3", sdd, True, relu=False)
                idn = a
                if p + ".downsample.0.weight" in sdd:
                    idn = bn_relu(conv(a, sdd[p + ".downsample.0.weight"], s_), p + ".downsample.1", sdd, True, relu=False)
                a = torch.relu(o + idn)
                rec.append((p, a))
        return rec
    A, B = run(False), run(True)
    prev = None
    for (p, a), (_, b) in zip(A, B):
        e = relc(b, a)
        m = a.mean((0, 2, 3)).abs(); s = a.std((0, 2, 3))
        print(f"N={N} {H}x{W} {p:22s} block output rel-L2 (centred) {e:.2e}  x{(e / prev if prev else 0):.2f}   std/|mean| median {float((s / (m + 1e-12)).median()):.2f}", flush=True)
        prev = e


with torch.no_grad():
    for shp in ((2, 72, 104), (2, 192, 192), (8, 96, 96)):
        growth(*shp)
