// unet.hip -- whole-network forward/backward sequencing (host code): the counterpart of
// UNet.forward (reference networks/unet_model.py:25-39) and of autograd's backward through it.
//
// Materialised between operators: the raw (pre-BatchNorm) conv outputs y_i and the ConvTranspose outputs u_j -- exactly
// what backward needs -- and the four pooled activations that feed the Down blocks (a quarter of their producers):
//   activated tensor  = loader(y_i, scale_i, shift_i, relu)         (BatchNorm + ReLU on load)
//   Down input        = pool_act(y_i, scale_i, shift_i)             (MaxPool2d of the activation, one streaming pass:
//                                                                     pooling on load cost the conv kernels more)
//   Up input          = loader(skip y_s) ++ loader(u_j, offset)     (pad + cat on load)
// A call may carry several forward passes of equal shape (desc.groups): BatchNorm statistics and constants per pass; a shorter
// pass may follow them (desc.tail: forward only -- the reference's low-quality-sample forward, train.py:740, whose logits nobody
// reads, rides behind the student's four gradient passes and moves the running statistics last, as it does there).
#include "common.h"
#include <string.h>

namespace ustrun {
namespace {

struct Plan {
    int N, C, H, W, K, base;
    int G, gN;                   // forward passes batched into this call, images per pass (BatchNorm is per pass)
    int L, Gb, ob;               // leading passes WITHOUT gradient (forward as any other; the backward starts behind them), the
                                 // passes the backward covers, the first image it covers
    int T, GP, Nb, pg, pgb;      // images of the shorter tail pass behind them (forward only, logits unused); passes incl. the tail;
                                 // G * gN = the images the head and the backward cover; pg = gN when the forward's batch has a pass
                                 // structure, pgb = the same for the backward's (which never sees the tail)
    int Hs[5], Ws[5];            // extent per level
    int cin[18], cout[18], lvl[18];
    int up_cin[4], up_cout[4];   // convT j: level of its input = 4-j (j = 0..3), output level 3-j
    int bil;                     // bilinear Up (unet_parts.py:48-51): no ConvTranspose -- u = the 2x-interpolated input, up_cout = up_cin
    int gi_head;                 // index of outc.conv.weight in model.parameters() order (62, bilinear: 54)
    int gi_conv(int i) const {   // index of conv i's weight in model.parameters() order; +1 gamma, +2 beta
        if (i < 10) return (i / 2) * 6 + (i % 2) * 3;
        const int j = (i - 10) / 2;
        return bil ? 30 + j * 6 + (i % 2) * 3 : 30 + j * 8 + 2 + (i % 2) * 3;
    }
    // BYTE offsets (activations are esz bytes per element, everything else f32)
    int esz;
    long y_off[18], aff_off[18]; // aff: scale, shift, mean, rstd (4*cout)
    long u_off[4];
    long pool_off[4];            // pooled activation feeding Down l+1 (input of conv 2l+2), kept for its weight gradient
    long act1_off[9];            // activation of conv 2k (the first of DoubleConv k) = the operand of conv 2k+1, written out by
                                 // ustrun_act16 on the levels from 256 channels (-1: applied on load)
    long act_off[4];             // un-pooled activation of conv 2l+1 = the decoder's skip operand at level l (-1: read through
                                 // BatchNorm + ReLU on load instead: f32 storage, odd extents, or USTRUN_DEBUG_FLAGS bit 24)
    long stat_off, tick_off, fwd_total;
    long wf_off[18], wd_off[18], uf_off[4], ud_off[4], pack_total;
    long da_off[18], du_off[4], dp_off[4], coef_off, part_off, bwd_total, part_bytes;
    long y_elems(int i) const { return (long)N * Hs[lvl[i]] * Ws[lvl[i]] * cout[i]; }
    long u_elems(int j) const { return (long)N * (2 * Hs[4 - j]) * (2 * Ws[4 - j]) * up_cout[j]; }
};

long align_up(long v, long a) { return (v + a - 1) / a * a; }

int make_plan(const ustrun_unet_desc_t* d, Plan& p) {
    USTRUN_CHECK(d, "unet: null descriptor");
    USTRUN_CHECK(d->N > 0 && d->C > 0 && d->K > 0 && d->K <= 8 && d->base >= 4 && d->base % 4 == 0,
                 "unet: bad N=%d C=%d K=%d base=%d", d->N, d->C, d->K, d->base);
    USTRUN_CHECK(d->H >= 16 && d->W >= 16, "unet: extent %dx%d too small for 4 poolings", d->H, d->W);
    USTRUN_CHECK(dtype_ok(d->dtype), "unet: dtype %d not built", d->dtype);
    p.N = d->N; p.C = d->C; p.H = d->H; p.W = d->W; p.K = d->K; p.base = d->base;
    p.G = d->groups > 1 ? d->groups : 1;
    p.T = d->tail;
    USTRUN_CHECK(p.T >= 0 && p.T < d->N && (d->N - p.T) % p.G == 0, "unet: N=%d is not groups=%d equal passes + a tail of %d", d->N, p.G, p.T);
    p.gN = (d->N - p.T) / p.G;
    USTRUN_CHECK(p.T < p.gN, "unet: the tail pass (%d images) must be shorter than the others (%d)", p.T, p.gN);
    p.L = d->lead;
    USTRUN_CHECK(p.L >= 0 && p.L < p.G, "unet: %d leading passes without gradient of %d passes", p.L, p.G);
    p.Gb = p.G - p.L; p.ob = p.L * p.gN;
    p.GP = p.G + (p.T > 0); p.Nb = p.Gb * p.gN; p.pg = p.GP > 1 ? p.gN : 0; p.pgb = p.Gb > 1 ? p.gN : 0;
    USTRUN_CHECK(p.GP <= 8, "unet: %d passes in one call (at most 8)", p.GP);
    p.Hs[0] = d->H; p.Ws[0] = d->W;
    for (int l = 1; l < 5; ++l) { p.Hs[l] = p.Hs[l - 1] / 2; p.Ws[l] = p.Ws[l - 1] / 2; }
    const int b = d->base;
    p.bil = d->bilinear ? 1 : 0;
    p.gi_head = p.bil ? 54 : 62;
    const int ch[5] = {b, 2 * b, 4 * b, 8 * b, p.bil ? 8 * b : 16 * b};       // (bilinear: Down(8b, 16b // 2), unet_model.py:17-18)
    for (int l = 0; l < 5; ++l) {                 // encoder double convs
        p.cin[2 * l] = l == 0 ? d->C : ch[l - 1]; p.cout[2 * l] = ch[l];
        p.cin[2 * l + 1] = ch[l]; p.cout[2 * l + 1] = ch[l];
        p.lvl[2 * l] = p.lvl[2 * l + 1] = l;
    }
    for (int j = 0; j < 4; ++j) {                 // decoder: up(j+1) works at level 3-j
        const int l = 3 - j;
        if (p.bil) {        // Up(in, out, bilinear) = Upsample + DoubleConv(in, out, in // 2) with out = in // 4 (up4: in // 2), unet_model.py:19-22
            const int x1ch = j == 0 ? ch[4] : p.cout[9 + 2 * j];
            p.up_cin[j] = p.up_cout[j] = x1ch;
            p.cin[10 + 2 * j] = ch[l] + x1ch; p.cout[10 + 2 * j] = p.cin[10 + 2 * j] / 2;
            p.cin[11 + 2 * j] = p.cout[10 + 2 * j]; p.cout[11 + 2 * j] = j < 3 ? ch[l] / 2 : ch[0];
        } else {
        p.up_cin[j] = ch[l + 1]; p.up_cout[j] = ch[l + 1] / 2;
        p.cin[10 + 2 * j] = ch[l] + p.up_cout[j]; p.cout[10 + 2 * j] = ch[l];
        p.cin[11 + 2 * j] = ch[l]; p.cout[11 + 2 * j] = ch[l];
        }
        p.lvl[10 + 2 * j] = p.lvl[11 + 2 * j] = l;
    }
    USTRUN_CHECK(!p.bil || b % 8 == 0, "unet: bilinear needs base %% 8 == 0 (base=%d)", b);
    p.esz = act_esz(d->dtype);
    const long E = p.esz;
    long o = 0;
    long stat_max = 0;
    // Tensor sizes here are powers of two (64 images x 256^2 x 64 channels x 2 B = 512 MB): two tensors a power of two apart
    // that one kernel streams in lockstep -- the two halves of a concat, dY in and dA out of an input gradient -- walk the
    // same HBM channels together (measured: 0.98 ms instead of 0.80 ms for the 128->64 concat conv).  Every tensor is
    // therefore followed by a gap that is a different odd multiple of 68 KB.
    int nplaced = 0;
    auto gap = [&]() { return (long)(2 * (nplaced++ % 8) + 1) * 69632; };
    for (int i = 0; i < 18; ++i) {
        p.y_off[i] = o; o = align_up(o + p.y_elems(i) * E, 256) + gap();
        p.aff_off[i] = o; o = align_up(o + 16L * p.cout[i] * p.GP, 256);      // [pass][scale, shift, mean, rstd]
        const long st = (long)ustrun_conv_mtiles(p.N, p.Hs[p.lvl[i]], p.Ws[p.lvl[i]], p.cout[i]) * 2 * p.cout[i];
        if (st > stat_max) stat_max = st;
    }
    for (int j = 0; j < 4; ++j) { p.u_off[j] = o; o = align_up(o + p.u_elems(j) * E, 256) + gap(); }
    for (int l = 0; l < 4; ++l) {
        p.pool_off[l] = o; o = align_up(o + (long)p.N * p.Hs[l + 1] * p.Ws[l + 1] * ch[l] * E, 256) + gap();
    }
    // The skip operands, materialised by the pool pass that reads the same tensor anyway (+2 B per element written) so that the
    // four concat convolutions and their weight gradients read PLAIN sources: applying BatchNorm + ReLU per staged item costs those
    // layers 14-19 % (tools/exp_cat_plain.py: 0.505 / 0.50 / 0.54 / 0.81 ms against 0.41 / 0.43 / 0.47 / 0.68 at N = 64), most on the
    // 64-column tile of up4.conv1, where a block amortises the transform over half the MFMAs
    for (int l = 0; l < 4; ++l) {
        // (the switch shapes the workspace the forward and the backward share, and the two may run on different threads -- autograd's
        // backward does: it is taken from the ENVIRONMENT's flags, a process-wide constant, never from the per-thread value)
        const bool on = E == 2 && !(p.Hs[l] & 1) && !(p.Ws[l] & 1) && !(env_debug_flags() & (1 << 24));
        p.act_off[l] = on ? o : -1;
        if (on) o = align_up(o + p.y_elems(2 * l + 1) * E, 256) + gap();
    }
    // ... and the operand of a DoubleConv's second convolution where a 4 B-per-element pass is cheaper than the transform it removes
    // from that convolution (14-19 %) and from its weight gradient (the activation is 1800-2300 of the 3200-3500 cycles a wave group
    // spends preparing a tile, 8.7): from 256 channels on (priced in DESIGN.md 9.13; 128 and 64 channels: the pass costs more)
    for (int k = 0; k < 9; ++k) {
        // (threshold measured on one box, 28.05-28.15 ms per step at 256: 128 -> 28.24, 64 -> 29.0, 512 -> 28.11)
        const bool on = E == 2 && p.cout[2 * k] >= 256 && p.cout[2 * k] % 8 == 0 && p.cout[2 * k] / 8 <= 256 && !(env_debug_flags() & (1 << 26));
        p.act1_off[k] = on ? o : -1;
        if (on) o = align_up(o + p.y_elems(2 * k) * E, 256) + gap();
    }
    p.stat_off = o; o = align_up(o + stat_max * 4, 256);
    p.tick_off = o; o += 256;              // BN_TICKETS counters of the one-launch statistics finalize (zeroed per forward)
    p.fwd_total = o;

    o = 0;
    for (int i = 0; i < 18; ++i) {
        // (covers f32 and the K-padded bf16 layout; USTRUN_F32X3: the f32 pack + its three bf16 planes = 2.5x, rounded up to 3x)
        const long n = 9L * align_up(p.cin[i], 8) * align_up(p.cout[i], 8) * (d->dtype == USTRUN_F32X3 ? 3 : 1);
        p.wf_off[i] = o; o = align_up(o + n, 64);
        p.wd_off[i] = o; o = align_up(o + n, 64);
    }
    for (int j = 0; j < 4; ++j) {
        const long n = p.bil ? 0 : 4L * align_up(p.up_cin[j], 8) * align_up(p.up_cout[j], 8) * (d->dtype == USTRUN_F32X3 ? 3 : 1);
        p.uf_off[j] = o; o = align_up(o + n, 64);
        p.ud_off[j] = o; o = align_up(o + n, 64);
    }
    p.pack_total = o;

    o = 34816;                                   // (the scratch starts off the workspace's phase as well)
    nplaced = 3;
    long part = ustrun_loss_partials_bytes(1, 1, 1);
    for (int i = 0; i < 18; ++i) {
        p.da_off[i] = o; o = align_up(o + p.y_elems(i) * E, 256) + gap();
        const long npix = (long)p.N * p.Hs[p.lvl[i]] * p.Ws[p.lvl[i]];
        long b1 = ustrun_bn_bwd_partials_bytes(npix, p.cout[i]);
        long b2 = ustrun_wgrad_partials_bytes(9, p.cin[i], p.cout[i], npix);
        if (b1 > part) part = b1;
        if (b2 > part) part = b2;
    }
    for (int j = 0; j < 4; ++j) {
        p.du_off[j] = o; o = align_up(o + p.u_elems(j) * E, 256) + gap();
        const int l = 3 - j;                                     // pooled grad of the skip at level l
        p.dp_off[j] = o; o = align_up(o + (long)p.N * p.Hs[l + 1] * p.Ws[l + 1] * ch[l] * E, 256) + gap();
        const long npix = (long)p.N * p.Hs[l + 1] * p.Ws[l + 1];
        long b2 = ustrun_wgrad_partials_bytes(4, p.up_cin[j], p.up_cout[j], npix);
        if (b2 > part) part = b2;
        long b3 = 512L * p.up_cout[j] * 4;
        if (b3 > part) part = b3;
    }
    long bh = 1024L * ((long)p.K * b + p.K) * 4;
    if (bh > part) part = bh;
    p.coef_off = o; o = align_up(o + 12L * 16 * b * p.G, 256);      // [pass][3][C]
    p.part_off = o; p.part_bytes = align_up(part, 256);
    o += p.part_bytes;
    p.bwd_total = o;
    return 0;
}

ustrun_src_t nhwc_src(const void* ptr, const float* aff, int C, int H, int W, int relu, int pool, int gN = 0) {
    ustrun_src_t s = {};
    s.ptr = ptr; s.scale = aff; s.shift = aff ? aff + C : nullptr;
    s.gN = gN; s.gstride = 4L * C;                  // pass g keeps its constants at aff + g * 4C
    s.C = C; s.H = H; s.W = W;
    s.sC = 1; s.sW = C; s.sH = (int64_t)W * C; s.sN = (int64_t)H * W * C;
    s.relu = relu; s.pool = pool;
    return s;
}

// sources of conv i (forward input), from the saved workspace
// bwd: the sources as the backward sees them -- its first image is p.ob (behind the leading passes), its pass structure p.pgb
int conv_sources_fwd(const Plan& p, const float* x, const char* ws, int i, ustrun_src_t* srcs);
int conv_sources(const Plan& p0, const float* x, const char* ws, int i, ustrun_src_t* srcs, bool bwd = false) {
    if (!bwd) return conv_sources_fwd(p0, x, ws, i, srcs);
    Plan p = p0;
    p.pg = p.pgb;
    const int ns = conv_sources_fwd(p, x, ws, i, srcs);
    for (int k = 0; k < ns; ++k) {
        ustrun_src_t& s = srcs[k];
        const int esz = (s.f32 || p.esz == 4) ? 4 : 2;
        s.ptr = (const char*)s.ptr + (int64_t)p.ob * s.sN * esz;
        if (s.scale) { s.scale += (int64_t)p.L * s.gstride; s.shift += (int64_t)p.L * s.gstride; }
    }
    return ns;
}
int conv_sources_fwd(const Plan& p, const float* x, const char* ws, int i, ustrun_src_t* srcs) {
    auto act = [&](int k, int pool) {
        return nhwc_src(ws + p.y_off[k], (const float*)(ws + p.aff_off[k]), p.cout[k], p.Hs[p.lvl[k]], p.Ws[p.lvl[k]], 1, pool,
                        p.pg);
    };
    auto act1 = [&](int i2) {       // operand of the second convolution i2 of a DoubleConv: written out, or through the transform
        const int k = (i2 - 1) / 2, c1 = i2 - 1;
        if (p.act1_off[k] >= 0)
            return nhwc_src(ws + p.act1_off[k], nullptr, p.cout[c1], p.Hs[p.lvl[c1]], p.Ws[p.lvl[c1]], 0, 0, p.pg);
        return act(c1, 0);
    };
    if (i == 0) {   // network input, NCHW
        ustrun_src_t s = {};
        s.ptr = x; s.C = p.C; s.H = p.H; s.W = p.W;
        s.sW = 1; s.sH = p.W; s.sC = (int64_t)p.H * p.W; s.sN = (int64_t)p.C * p.H * p.W;
        s.f32 = 1;                                  // the network input is f32 whatever the storage dtype
        srcs[0] = s;
        return 1;
    }
    if (i < 10) {
        if (i % 2 == 1) { srcs[0] = act1(i); return 1; }
        const int l = i / 2;       // Down l: the pooled activation was materialised by ustrun_pool_act (plain tensor)
        srcs[0] = nhwc_src(ws + p.pool_off[l - 1], nullptr, p.cout[i - 1], p.Hs[l], p.Ws[l], 0, 0, p.pg);
        return 1;
    }
    if (i % 2 == 1) { srcs[0] = act1(i); return 1; }
    const int j = (i - 10) / 2, l = 3 - j;
    const int skip = 2 * l + 1;
    srcs[0] = p.act_off[l] >= 0 ? nhwc_src(ws + p.act_off[l], nullptr, p.cout[skip], p.Hs[l], p.Ws[l], 0, 0, p.pg)
                                : act(skip, 0);
    ustrun_src_t u = nhwc_src(ws + p.u_off[j], nullptr, p.up_cout[j], 2 * p.Hs[l + 1], 2 * p.Ws[l + 1], 0, 0);
    u.off_y = (p.Hs[l] - 2 * p.Hs[l + 1]) / 2;       // F.pad(diff//2, ...) of the reference
    u.off_x = (p.Ws[l] - 2 * p.Ws[l + 1]) / 2;
    srcs[1] = u;
    return 2;
}

}  // namespace
}  // namespace ustrun

using namespace ustrun;

extern "C" int64_t ustrun_unet_packed_bytes(const ustrun_unet_desc_t* d) {
    Plan p; if (make_plan(d, p)) return -1;
    return p.pack_total * 4;
}
extern "C" int64_t ustrun_unet_fwd_workspace_bytes(const ustrun_unet_desc_t* d) {
    Plan p; if (make_plan(d, p)) return -1;
    return p.fwd_total;
}
extern "C" int64_t ustrun_unet_bwd_scratch_bytes(const ustrun_unet_desc_t* d) {
    Plan p; if (make_plan(d, p)) return -1;
    return p.bwd_total;
}

extern "C" int ustrun_unet_pack(const ustrun_unet_desc_t* d, ustrun_stream_t s) {
    Plan p; USTRUN_TRY(make_plan(d, p));
    USTRUN_CHECK(d->packed, "unet_pack: packed arena missing");
    float* pk = (float*)d->packed;
    if (d->dtype == USTRUN_D16) {                 // every layer in one launch
        PackJobs jobs;
        for (int i = 0; i < 18; ++i) {
            USTRUN_CHECK(d->conv_w[i], "unet_pack: conv weight %d missing", i);
            jobs.j[i] = PackJob{d->conv_w[i], pk + p.wf_off[i], pk + p.wd_off[i], p.cout[i], p.cin[i], 9, 0};
        }
        for (int j = 0; j < 4 && !p.bil; ++j) {
            USTRUN_CHECK(d->up_w[j], "unet_pack: up weight %d missing", j);
            jobs.j[18 + j] = PackJob{d->up_w[j], pk + p.uf_off[j], pk + p.ud_off[j], p.up_cout[j], p.up_cin[j], 4, 1};
        }
        return pack_bf16_multi(jobs, p.bil ? 18 : 22, (hipStream_t)s);
    }
    for (int i = 0; i < 18; ++i) {
        USTRUN_CHECK(d->conv_w[i], "unet_pack: conv weight %d missing", i);
        USTRUN_TRY(ustrun_pack_conv3x3(d->conv_w[i], p.cout[i], p.cin[i], pk + p.wf_off[i], pk + p.wd_off[i], d->dtype, s));
    }
    for (int j = 0; j < 4 && !p.bil; ++j) {
        USTRUN_CHECK(d->up_w[j], "unet_pack: up weight %d missing", j);
        USTRUN_TRY(ustrun_pack_convT2x2(d->up_w[j], p.up_cin[j], p.up_cout[j], pk + p.uf_off[j], pk + p.ud_off[j], d->dtype, s));
    }
    return 0;
}

extern "C" int ustrun_unet_forward(const ustrun_unet_desc_t* d, const float* x, float* logits, float* feat,
                                   void* workspace, ustrun_stream_t s) {
    Plan p; USTRUN_TRY(make_plan(d, p));
    ShortLastPass declared_tail(p.T > 0);
    USTRUN_CHECK(x && logits && workspace && d->packed, "unet_forward: null pointer");
    char* ws = (char*)workspace;
    const float* pk = (const float*)d->packed;
    float* stat = (float*)(ws + p.stat_off);
    auto affp = [&](int k) { return (float*)(ws + p.aff_off[k]); };
    if (!d->train) {     // eval: every layer's scale/shift from the running statistics, one launch for all 18 layers
        float* affs[18];
        for (int i = 0; i < 18; ++i) affs[i] = affp(i);
        USTRUN_TRY(bn_eval_affine_layers(18, p.cout, d->bn_w, d->bn_b, (const float* const*)d->bn_rm, (const float* const*)d->bn_rv, affs,
                                         d->eps, p.GP, (hipStream_t)s));
    }
    unsigned* tickets = (unsigned*)(ws + p.tick_off);
    static_assert(BN_TICKETS * sizeof(unsigned) <= 256, "ticket area");
    // (one-launch statistics finalize, ustrun_debug_flags bit 22: bit-identical, and measured a wash -- 29.80 vs 29.65 ms per step
    // on one box, profiles/r04_ab_bn_fused_finalize.log: back-to-back small launches overlap their launch latency, the fused
    // kernel serialises a last-block tail behind its tickets -- so the two launches stay the default)
    if (d->train && (g_debug_flags & 4194304)) {
        const hipError_t e = hipMemsetAsync(tickets, 0, BN_TICKETS * sizeof(unsigned), (hipStream_t)s);
        USTRUN_CHECK(e == hipSuccess, "unet_forward: ticket reset: %s", hipGetErrorString(e));
    }
    for (int i = 0; i < 18; ++i) {
        if (i >= 10 && i % 2 == 0) {   // Up: ConvTranspose of the previous level's output first
            const int j = (i - 10) / 2, l = 3 - j;
            const int prev = (j == 0) ? 9 : i - 1;
            ustrun_src_t a = nhwc_src(ws + p.y_off[prev], affp(prev), p.cout[prev], p.Hs[l + 1], p.Ws[l + 1], 1, 0,
                                      p.pg);
            prof_set_tag(20 + j, p.N);
            if (p.bil) USTRUN_TRY(ustrun_upsample2x_act(&a, p.N, ws + p.u_off[j], d->dtype == USTRUN_F32X3 ? USTRUN_F32 : d->dtype, s));
            else
            USTRUN_TRY(ustrun_convT2x2_fwd(&a, pk + p.uf_off[j], d->up_b[j], p.N, p.Hs[l + 1], p.Ws[l + 1], p.up_cout[j],
                                           ws + p.u_off[j], d->dtype, s));
        }
        if (i >= 2 && i < 10 && i % 2 == 0) {      // MaxPool2d of the previous level's activated output
            const int l = i / 2;
            ustrun_src_t a = nhwc_src(ws + p.y_off[i - 1], affp(i - 1), p.cout[i - 1], p.Hs[l - 1], p.Ws[l - 1], 1, 0,
                                      p.pg);
            USTRUN_TRY(ustrun_pool_act2(&a, p.N, ws + p.pool_off[l - 1], p.act_off[l - 1] >= 0 ? ws + p.act_off[l - 1] : nullptr,
                                        d->dtype, s));
        }
        ustrun_src_t srcs[2];
        const int ns = conv_sources(p, x, ws, i, srcs);
        const int H = p.Hs[p.lvl[i]], W = p.Ws[p.lvl[i]];
        int stat_rows = 0;
        prof_set_tag(i, p.N);
        int tail_rows = 0;           // statistics rows of the tail pass (the last ones)
        if (i == 0 && d->train && p.GP > 1 && !conv_first_supported(srcs[0], p.cout[0])) {
            // the network input carries no pass structure, and the generic kernels' statistics rows are 128-pixel runs of the
            // whole batch (the first-convolution kernels' rows never cross an image): one launch per pass keeps every row
            // inside its pass (base widths other than 64 only)
            for (int g = 0; g < p.GP; ++g) {
                ustrun_src_t sg = srcs[0];
                sg.ptr = (const float*)srcs[0].ptr + (long)g * p.gN * srcs[0].sN;
                int rows = 0;
                USTRUN_TRY(ustrun_conv3x3_fwd_rows(&sg, 1, pk + p.wf_off[0], g < p.G ? p.gN : p.T, H, W, p.cout[0],
                                                   ws + p.y_off[0] + (long)g * p.gN * H * W * p.cout[0] * p.esz,
                                                   stat + (long)stat_rows * 2 * p.cout[0], &rows, d->dtype, s));
                stat_rows += rows;
                if (g >= p.G) tail_rows = rows;
            }
        } else {
            USTRUN_TRY(ustrun_conv3x3_fwd_rows(srcs, ns, pk + p.wf_off[i], p.N, H, W, p.cout[i], ws + p.y_off[i],
                                               d->train ? stat : nullptr, &stat_rows, d->dtype, s));
            if (p.T > 0 && d->train) {       // one launch per pass: the last pass's rows; one launch: every image the same number
                tail_rows = conv_last_pass_rows();
                if (tail_rows < 0) {
                    USTRUN_CHECK(stat_rows % p.N == 0, "unet_forward: %d statistics rows over %d images", stat_rows, p.N);
                    tail_rows = stat_rows / p.N * p.T;
                }
            }
        }
        float* aff = affp(i);
        const int C = p.cout[i];
        if (d->train) {     // statistics per pass, running buffers updated pass after pass as separate calls would
            USTRUN_CHECK((stat_rows - tail_rows) % p.G == 0 && tail_rows >= 0 && (tail_rows > 0) == (p.T > 0),
                         "unet_forward: %d statistics rows (%d of the tail) do not split into %d passes", stat_rows, tail_rows, p.G);
            const int rpg = (stat_rows - tail_rows) / p.G;
            USTRUN_TRY(bn_finalize_passes(stat, rpg, p.G, C, (int64_t)p.gN * H * W, d->bn_w[i], d->bn_b[i], d->bn_rm[i],
                                          d->bn_rv[i], d->bn_nbt[i], d->momentum, d->eps, d->update_running, aff, aff + C,
                                          aff + 2 * C, aff + 3 * C, 4L * C, (hipStream_t)s, (g_debug_flags & 4194304) ? tickets : nullptr,
                                          tail_rows, (int64_t)p.T * H * W));
        }
        if (i % 2 == 0 && p.act1_off[i / 2] >= 0) {        // the second convolution's operand, written out (see make_plan)
            ustrun_src_t a = nhwc_src(ws + p.y_off[i], aff, C, H, W, 1, 0, p.pg);
            USTRUN_TRY(ustrun_act16(&a, p.N, ws + p.act1_off[i / 2], d->dtype, s));
        }
    }
    prof_set_tag(-1, 0);
    const int C = p.cout[17];
    const long gpix = (long)p.gN * p.H * p.W;
    for (int g = 0; g < p.G; ++g) {
        const float* ag = affp(17) + 4L * C * g;
        const char* yg = ws + p.y_off[17] + g * gpix * C * p.esz;
        USTRUN_TRY(ustrun_head_fwd(yg, ag, ag + C, gpix, p.H * p.W, C, p.K, d->head_w, d->head_b, logits + g * gpix * p.K,
                                   d->dtype, s));
        if (feat)
            USTRUN_TRY(ustrun_bn_relu_apply(yg, ag, ag + C, gpix, C, p.H * p.W, feat + g * gpix * C, 1, d->dtype, s));
    }
    return 0;
}

extern "C" int ustrun_unet_backward(const ustrun_unet_desc_t* d, const float* x, const float* dlogits, void* workspace,
                                    void* scratch, float* const* grads, int accumulate, ustrun_stream_t s) {
    return ustrun_unet_backward_part(d, x, dlogits, workspace, scratch, grads, accumulate, 0, s);
}

// part 0: everything; part 1: head + decoder (the gradients of up1..up4 and outc, the contiguous tail of the parameter
// order, are final afterwards); part 2: encoder, continuing from the same scratch -- or, finer, part 3: down4 (its
// gradients are final afterwards) then part 4: down3..inc.  Lets the caller start the all-reduce of the decoder gradients
// while the encoder half still runs, and that of down4's while the high-resolution encoder layers run.
extern "C" int ustrun_unet_backward_part(const ustrun_unet_desc_t* d, const float* x, const float* dlogits, void* workspace,
                                         void* scratch, float* const* grads, int accumulate, int which, ustrun_stream_t s) {
    USTRUN_CHECK(which >= 0 && which <= 4, "unet_backward: part %d", which);
    Plan p; USTRUN_TRY(make_plan(d, p));
    ShortLastPass declared_tail(p.T > 0);
    USTRUN_CHECK(x && dlogits && workspace && scratch && grads && d->packed, "unet_backward: null pointer");
    USTRUN_CHECK(d->train, "unet_backward: forward must have run in train mode");
    const char* ws = (const char*)workspace;
    char* sc = (char*)scratch;
    const float* pk = (const float*)d->packed;
    float* coef = (float*)(sc + p.coef_off);
    float* part = (float*)(sc + p.part_off);
    // (everything the backward reads starts behind the leading passes: image p.ob of every tensor, pass p.L of every constant table)
    auto affp = [&](int k) { return (const float*)(ws + p.aff_off[k]) + 4L * p.cout[k] * p.L; };
    auto yb = [&](int k) { return ws + p.y_off[k] + (long)p.ob * p.Hs[p.lvl[k]] * p.Ws[p.lvl[k]] * p.cout[k] * p.esz; };
    dlogits += (long)p.ob * p.K * p.H * p.W;
    const int dt = d->dtype;

    int head_bn_rows = 0;       // > 0: the head kernel also formed layer 17's BatchNorm-backward sums (rows per pass, in `part`)
    if (which <= 1) {   // head: all passes in one launch (blockIdx.y = pass: its BatchNorm constants on load)
        const int C = p.cout[17];
        const long gpix = (long)p.gN * p.H * p.W;
        USTRUN_TRY(head_bwd_passes(dlogits, yb(17), affp(17), affp(17) + C, gpix, p.H * p.W, C, p.K, d->head_w,
                                   sc + p.da_off[17], grads[p.gi_head], grads[p.gi_head + 1], accumulate, part, p.part_bytes, dt, p.Gb, 4L * C,
                                   (hipStream_t)s, (g_debug_flags & 8388608) ? nullptr : &head_bn_rows));
    }
    // layers 17..10 = decoder, 9..8 = down4 (57 of the encoder's 75 MB of gradients, and the first to finish), 7..0 = the rest
    const int i_hi = which <= 1 ? 17 : (which == 4 ? 7 : 9);
    const int i_lo = which == 1 ? 10 : (which == 3 ? 8 : 0);
    int dgrad_bn_rows = 0;      // > 0: the input gradient that wrote this layer's da also formed its BatchNorm-backward sums (rows in `part`)
    for (int i = i_hi; i >= i_lo; --i) {
        if (i == ((g_debug_flags >> 16) & 31) - 1) return 0;      // debugging aid (ustrun_debug_flags bits 16-20 = layer + 1): stop before this layer

        const int l = p.lvl[i], H = p.Hs[l], W = p.Ws[l], C = p.cout[i];
        const float* aff = affp(i);
        const int gi = p.gi_conv(i);
        // the encoder outputs x1..x4 (convs 1,3,5,7) also feed a MaxPool: add the routed pooled grad
        const bool pooled = (i < 8) && (i % 2 == 1);
        const void* dp = pooled ? sc + p.dp_off[3 - l] : nullptr;
        void* da = sc + p.da_off[i];
        {   // BatchNorm backward is a per-pass reduction: all passes in one launch per kernel (blockIdx.y = pass)
            const long act = (long)p.gN * H * W * C, pl = (long)p.gN * (H / 2) * (W / 2) * C;
            if (i == 17 && head_bn_rows > 0) {      // the sums came with the head's partial rows: no reduce pass over da and y
                const long row = (long)p.K * C + p.K + 2L * C;
                USTRUN_TRY(bn_bwd_finalize_rows(part, head_bn_rows, row, (long)p.K * C + p.K, C, (int64_t)p.gN * H * W, d->bn_w[i],
                                                aff + 2 * C, aff + 3 * C, grads[gi + 1], grads[gi + 2], accumulate, coef, p.Gb,
                                                4L * C, (hipStream_t)s));
            } else if (dgrad_bn_rows > 0) {         // ... or with the rows the producing input gradient wrote (below)
                USTRUN_CHECK(dgrad_bn_rows % p.Gb == 0, "unet_backward: %d sum rows do not split into %d passes", dgrad_bn_rows, p.Gb);
                USTRUN_TRY(bn_bwd_finalize_stat(part, dgrad_bn_rows / p.Gb, p.Gb, C, (int64_t)p.gN * H * W, d->bn_w[i], aff + 2 * C,
                                                aff + 3 * C, 4L * C, grads[gi + 1], grads[gi + 2], accumulate, coef, (hipStream_t)s));
            } else
            USTRUN_TRY(bn_bwd_reduce_passes(da, dp, yb(i), aff, aff + C, aff + 2 * C, aff + 3 * C, d->bn_w[i], p.gN, H, W,
                                            C, grads[gi + 1], grads[gi + 2], accumulate, coef, part, p.part_bytes, dt, p.Gb, act,
                                            pl, 4L * C, (hipStream_t)s));
            USTRUN_TRY(bn_bwd_apply_passes(da, dp, yb(i), aff, aff + C, coef, p.gN, H, W, C, da, dt, p.Gb, act, pl,
                                           4L * C, (hipStream_t)s));
        }
        dgrad_bn_rows = 0;
        ustrun_src_t srcs[2];
        const int ns = conv_sources(p, x, ws, i, srcs, true);
        prof_set_tag(200 + i, p.Nb);
        USTRUN_TRY(ustrun_conv3x3_wgrad(srcs, ns, da, p.Nb, H, W, C, grads[gi], accumulate, part, p.part_bytes, dt, s));
        prof_set_tag(-1, 0);
        if (i == 0) break;
        const float* wd = pk + p.wd_off[i];
        struct Untag { ~Untag() { prof_set_tag(-1, 0); } } untag_;
        prof_set_tag(100 + i, p.Nb);
        if (i < 10 && i % 2 == 0) {            // Down conv: grad wrt the pooled activation of conv i-1
            USTRUN_TRY(ustrun_conv3x3_dgrad(da, wd, p.Nb, H, W, C, p.cin[i], sc + p.dp_off[3 - (l - 1)], p.cin[i], nullptr, 0,
                                            0, 0, 0, dt, s));
        } else if (i >= 10 && i % 2 == 0) {    // Up conv: split into the skip grad and the ConvTranspose-output grad
            const int j = (i - 10) / 2, skip = 2 * l + 1;
            const int uh = 2 * p.Hs[l + 1], uw = 2 * p.Ws[l + 1];
            USTRUN_TRY(ustrun_conv3x3_dgrad(da, wd, p.Nb, H, W, C, p.cin[i], sc + p.da_off[skip], p.cout[skip],
                                            sc + p.du_off[j], uh, uw, (H - uh) / 2, (W - uw) / 2, dt, s));
            const int prev = (j == 0) ? 9 : i - 1;
            if (p.bil) {                       // the interpolation's adjoint: du -> da of the block below (no parameters)
                prof_set_tag(120 + j, p.Nb);
                USTRUN_TRY(ustrun_upsample2x_bwd_t(sc + p.du_off[j], p.Nb, p.Hs[l + 1], p.Ws[l + 1], p.up_cout[j], sc + p.da_off[prev],
                                                   dt == USTRUN_F32X3 ? USTRUN_F32 : dt, s));
                continue;
            }
            ustrun_src_t a = nhwc_src(yb(prev), affp(prev), p.cout[prev], p.Hs[l + 1], p.Ws[l + 1], 1, 0,
                                      p.pgb);
            const int ub = 30 + j * 8;
            prof_set_tag(220 + j, p.Nb);
            USTRUN_TRY(ustrun_convT2x2_wgrad(&a, sc + p.du_off[j], p.Nb, p.Hs[l + 1], p.Ws[l + 1], p.up_cout[j], grads[ub],
                                             grads[ub + 1], accumulate, part, p.part_bytes, dt, s));
            prof_set_tag(120 + j, p.Nb);
            // (da of the conv2 one level down: its BatchNorm-backward sums ride along where the layer is handled by this same call)
            const float* pa = affp(prev);
            const int Cp = p.up_cin[j];
            // (not for down4's second BatchNorm, j = 0: the backward may be split right there -- parts 1 | 2 -- and the split and
            // the un-split call must sum in the same order: test_backward_in_two_parts_equals_one_call)
            if (!(g_debug_flags & (1 << 25)) && j > 0 && prev >= i_lo &&
                (long)ustrun_conv_mtiles(p.Nb, p.Hs[l + 1], p.Ws[l + 1], Cp) * 2 * Cp * 4 <= p.part_bytes)
                USTRUN_TRY(ustrun_convT2x2_dgrad_bnsum(sc + p.du_off[j], pk + p.ud_off[j], p.Nb, p.Hs[l + 1], p.Ws[l + 1], p.up_cout[j], Cp,
                                                       sc + p.da_off[prev], yb(prev), pa, pa + Cp, p.pgb, 4L * Cp,
                                                       part, &dgrad_bn_rows, dt, s));
            if (dgrad_bn_rows == 0)
            USTRUN_TRY(ustrun_convT2x2_dgrad(sc + p.du_off[j], pk + p.ud_off[j], p.Nb, p.Hs[l + 1], p.Ws[l + 1], p.up_cout[j],
                                             p.up_cin[j], sc + p.da_off[prev], dt, s));
        } else {                               // second conv of a DoubleConv
            // its input gradient IS da of the BatchNorm + ReLU between the two convolutions: where the halo kernel's fused epilogue
            // covers the shape (16-bit storage, >= 128 channels, the 256-pixel tiles; ustrun_debug_flags bit 25: never) the launch
            // also forms that layer's backward sums -- rows in `part`, which nobody touches before the next iteration reads them
            const float* pa = affp(i - 1);
            const int Cp = p.cin[i];
            const long need = (long)ustrun_conv_mtiles(p.Nb, H, W, Cp) * 2 * Cp * 4;
            if (!(g_debug_flags & (1 << 25)) && i - 1 >= i_lo && need <= p.part_bytes)
                USTRUN_TRY(ustrun_conv3x3_dgrad_bnsum(da, wd, p.Nb, H, W, C, Cp, sc + p.da_off[i - 1], yb(i - 1), pa, pa + Cp,
                                                      p.pgb, 4L * Cp, part, &dgrad_bn_rows, dt, s));
            if (dgrad_bn_rows == 0)
                USTRUN_TRY(ustrun_conv3x3_dgrad(da, wd, p.Nb, H, W, C, Cp, sc + p.da_off[i - 1], Cp, nullptr, 0, 0, 0, 0, dt, s));
        }
    }
    return 0;
}
