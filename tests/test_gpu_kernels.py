"""Per-kernel parity on the MI355X: each C-ABI operator against the CPU oracle's restatement."""
import math

import numpy as np
import pytest
import torch

import boficap_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    from boficap_amd import hip
    assert torch.cuda.is_available(), "these tests need the MI355X"
    hip.lib()
    return hip


def _rng(seed):
    return torch.Generator().manual_seed(seed)


def _bf(x):
    return x.to(torch.bfloat16).float()


@pytest.mark.parametrize("d", [128, 512, 2048])
@pytest.mark.parametrize("out", ["f32", "bf16"])
def test_layernorm(H, d, out):
    g = _rng(1)
    x = torch.randn(77, d, generator=g) * 3 + 0.5
    a, b = torch.rand(d, generator=g) + 0.5, torch.randn(d, generator=g) * 0.1
    ref = O.layer_norm(x, {"n.a_2": a, "n.b_2": b}, "n")
    y = torch.empty(77, d, dtype=torch.float32 if out == "f32" else torch.bfloat16, device="cuda")
    xc, ac, bc = x.cuda(), a.cuda(), b.cuda()            # keep the device copies alive across the launch
    H.check(H.lib().bofi_layernorm(H.ptr(xc), H.ptr(ac), H.ptr(bc), H.ptr(y), H.dtype_code(y), 77, d, H.stream_ptr()))
    torch.cuda.synchronize()
    if out == "f32":
        assert (y.cpu() - ref).abs().max() < 2e-5
    else:
        assert torch.equal(y.cpu().float(), _bf(ref)) or (y.cpu().float() - _bf(ref)).abs().max() < 4e-2  # 1 bf16 ulp at |x|<8


def _linear(H, x, w, bias, residual, y_dtype, relu=0, row_len=None, rpg=0, ldr=None):
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty(M, N, dtype=y_dtype, device="cuda")
    H.check(H.lib().bofi_linear(H.ptr(x), H.dtype_code(x), K, H.ptr(w), H.dtype_code(w), H.ptr(bias), H.ptr(residual),
                                (N if ldr is None else ldr), H.ptr(y), H.dtype_code(y), N, M, N, K, relu, H.ptr(row_len), rpg, H.stream_ptr()))
    torch.cuda.synchronize()
    return y


@pytest.mark.parametrize("M,N,K", [(2304, 512, 2048), (1280, 1536, 512), (64, 512, 512), (64, 2048, 512), (130, 9491, 512),
                                   (1, 512, 512), (300, 100, 128), (72, 64, 64)])
def test_linear_f32(H, M, N, K):
    g = _rng(M + N + K)
    x, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K)
    b, r = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    ref = r + torch.relu(torch.nn.functional.linear(x.double(), w.double(), b.double())).float()
    y = _linear(H, x.cuda(), w.cuda(), b.cuda(), r.cuda(), torch.float32, relu=1)
    assert (y.cpu() - ref).abs().max() < 2e-5 * math.sqrt(K)
    # exact-zero rows past each group's length, no bias/residual
    lens = torch.tensor([(i * 7 + 5) % 37 for i in range((M + 35) // 36)], dtype=torch.int32)
    y = _linear(H, x.cuda(), w.cuda(), None, None, torch.float32, row_len=lens.cuda(), rpg=36)
    ref = torch.nn.functional.linear(x, w)
    keep = (torch.arange(M) % 36) < lens[torch.arange(M) // 36]
    assert (y.cpu()[~keep] == 0).all()
    assert (y.cpu()[keep] - ref[keep]).abs().max() < 2e-5 * math.sqrt(K)


@pytest.mark.parametrize("M,N,K", [(2304, 512, 2048), (1280, 2048, 512), (64, 512, 2048), (130, 9491, 512), (50, 96, 128)])
@pytest.mark.parametrize("xdt", ["bf16", "f32"])
def test_linear_bf16(H, M, N, K, xdt):
    g = _rng(M * 3 + N + K)
    x, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K)
    b, r = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    ref = (r.double() + torch.nn.functional.linear(_bf(x).double(), _bf(w).double(), b.double())).float()
    xin = x.cuda().to(torch.bfloat16) if xdt == "bf16" else x.cuda()
    y = _linear(H, xin, w.cuda().to(torch.bfloat16), b.cuda(), r.cuda(), torch.float32)
    assert (y.cpu() - ref).abs().max() < 1e-4 * math.sqrt(K)          # fp32 accumulation of exact bf16 products
    yb = _linear(H, xin, w.cuda().to(torch.bfloat16), b.cuda(), None, torch.bfloat16, relu=1)
    refb = torch.relu(ref - r)
    assert (yb.cpu().float() - refb).abs().max() < 0.02 * max(1.0, float(refb.abs().max()))


def test_linear_broadcast_residual(H):
    g = _rng(5)
    x, w, r = torch.randn(64, 512, generator=g), torch.randn(512, 512, generator=g) / 22, torch.randn(512, generator=g)
    y = _linear(H, x.cuda(), w.cuda(), None, r.cuda(), torch.float32, ldr=0)
    assert (y.cpu() - (r[None] + x @ w.T)).abs().max() < 1e-3


def _attention_ref(q, k, v, klen, h):
    """oracle attention() core on already-projected q/k/v, dense mask from per-row prefix lengths"""
    B, Lq, d = q.shape
    Lk = k.shape[1]
    mask = torch.arange(Lk)[None, None, :] < klen[:, :, None]
    qh, kh, vh = (t.view(B, -1, h, 64).transpose(1, 2) for t in (q, k, v))
    s = torch.matmul(qh, kh.transpose(-2, -1)) / math.sqrt(64)
    s = s.masked_fill(mask.unsqueeze(1) == 0, float("-inf"))
    return torch.matmul(torch.softmax(s, -1), vh).transpose(1, 2).reshape(B, Lq, d)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("B,h,Lq,Lk", [(5, 8, 36, 36), (3, 8, 20, 20), (4, 2, 22, 22), (6, 8, 20, 36), (7, 8, 1, 36), (2, 8, 100, 100), (2, 4, 20, 90), (3, 8, 1, 100), (2, 8, 40, 128), (2, 8, 33, 70)])
def test_attention(H, dt, B, h, Lq, Lk):
    g = _rng(B * 100 + Lq + Lk)
    d = h * 64
    q, k, v = (torch.randn(B, L, d, generator=g) for L in (Lq, Lk, Lk))
    klen = torch.randint(1, Lk + 1, (B, Lq), generator=g).to(torch.int32)
    klen[0, 0] = 0                                     # fully masked row -> NaN like softmax(all -inf)
    tdt = torch.float32 if dt == "f32" else torch.bfloat16
    if dt == "bf16":
        q, k, v = _bf(q), _bf(k), _bf(v)
    ref = _attention_ref(q, k, v, klen.long(), h)
    out = torch.empty(B, Lq, d, dtype=tdt, device="cuda")
    qc, kc, vc = (t.cuda().to(tdt).contiguous() for t in (q, k, v))
    klc = klen.cuda()
    H.check(H.lib().bofi_attention(H.ptr(qc), d, H.ptr(kc), d, H.ptr(vc), d, H.ptr(out), d, H.dtype_code(out), B, h, Lq, Lk,
                                   H.ptr(klc), Lq, 1, H.stream_ptr()))
    torch.cuda.synchronize()
    o = out.cpu().float()
    assert torch.isnan(o[0, 0]).all() and torch.isnan(ref[0, 0]).all()
    o[0, 0], ref[0, 0] = 0, 0
    assert not torch.isnan(o).any()
    assert (o - ref).abs().max() < (2e-5 if dt == "f32" else 3e-2)


def test_vocab_finalize(H):
    g = _rng(9)
    S, V, B = 20, 9491, 3
    lg = torch.randn(B * S, V, generator=g) * 3
    lg[5, 100] = lg[5, 7000] = lg[5].max() + 1.0           # tie: the lower index must win
    lg[7, 33] = float("nan")                                 # NaN row: index 0 after log_softmax
    lg[9, :] = float("-inf"); lg[9, 1234] = 0.5              # one finite entry
    ntok = torch.tensor([20, 4, 0], dtype=torch.int32)
    ref_lp = torch.log_softmax(lg, dim=1)
    ref_id = torch.max(ref_lp, 1)[1]
    for b in range(B):
        ref_id[b * S + int(ntok[b]):(b + 1) * S] = 0
    x = lg.clone().cuda()
    seq = torch.empty(B * S, dtype=torch.int64, device="cuda")
    ntc = ntok.cuda()
    H.check(H.lib().bofi_vocab_finalize(H.ptr(x), B * S, V, S, 1, H.ptr(ntc), 0, H.ptr(seq), H.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(seq.cpu(), ref_id)
    assert ref_id[5] == 100 and ref_id[7] == 0
    got = x.cpu()
    assert torch.isnan(got[7]).all()
    m = ~torch.isnan(ref_lp) & ~torch.isinf(ref_lp)
    assert (got[m] - ref_lp[m]).abs().max() < 2e-5
    assert (got[9][torch.isinf(ref_lp[9])] == float("-inf")).all()
    # raw-logit mode: values untouched, first NaN index returned
    x = lg.clone().cuda()
    H.check(H.lib().bofi_vocab_finalize(H.ptr(x), B * S, V, S, 0, None, 0, H.ptr(seq), H.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(x.cpu()[~torch.isnan(lg)], lg[~torch.isnan(lg)])
    assert seq[7].item() == 33 and seq[5].item() == 100


def _row_stats(x32, groups):
    """partial (sum, sum of squares) per row and column group, as the producers write them: [M][groups][2]"""
    M, d = x32.shape
    v = x32.view(M, groups, d // groups)
    return torch.stack([v.sum(-1), (v * v).sum(-1)], -1).contiguous()


@pytest.mark.parametrize("M", [1, 17, 64, 65, 256, 320])
@pytest.mark.parametrize("mode", ["plain", "ln_relu", "res_stats", "splitk"])
def test_rowgemm(H, M, mode):
    """bofi_rowgemm (bound_ops.hip): y = epilogue(x w^T) for a few rows, against float64 on the bf16-rounded operands: LayerNorm fold
    from row statistics (16- and 32-column groups), ReLU, float32 residual, bf16 copy, output row statistics, split-K slabs; rows
    that do not fill a 16-row MFMA group or a 64-row block."""
    g = _rng(3 + M)
    K = 2048 if mode == "splitk" else 512
    N = 512 if mode in ("res_stats", "splitk") else 2048
    splitk = K // 512
    x32 = torch.randn(M, K, generator=g) * 1.5 + 0.2
    x = x32.to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, generator=g) * 0.1
    res = torch.randn(M, N, generator=g)
    acc = x.double() @ w.double().t()
    kw = dict(stats=None, groups=0, colsum=None, residual=None, relu=0, y=None, yb=None, stats_out=None)
    if mode == "plain":
        ref = acc + bias.double()
        kw.update(y=torch.empty(M, N, device="cuda"))
    elif mode == "ln_relu":
        groups = 32 if M % 2 else 16
        mean = x32.double().mean(1, keepdim=True)
        rstd = 1.0 / (x32.double().std(1, keepdim=True) + 1e-6)
        colsum = w.double().sum(1)
        ref = torch.relu(rstd * (acc - mean * colsum) + bias.double())
        kw.update(stats=_row_stats(x32, groups).cuda(), groups=groups, colsum=colsum.float().cuda(), relu=1,
                  yb=torch.empty(M, N, dtype=torch.bfloat16, device="cuda"))
    elif mode == "res_stats":
        ref = acc + bias.double() + res.double()
        kw.update(residual=res.cuda(), y=torch.empty(M, N, device="cuda"), yb=torch.empty(M, N, dtype=torch.bfloat16, device="cuda"),
                  stats_out=torch.zeros(M, N // 16, 2, device="cuda"))
    else:
        ref = acc + bias.double() + res.double()
        kw.update(residual=res.cuda(), y=torch.empty(splitk, M, N, device="cuda"))
    xc, wc, bc = x.cuda(), w.cuda(), bias.cuda()
    H.check(H.lib().bofi_rowgemm(H.ptr(xc), K, H.ptr(wc), H.ptr(bc), H.ptr(kw["stats"]), kw["groups"], H.ptr(kw["colsum"]), H.ptr(kw["residual"]), N,
                                 H.ptr(kw["y"]), N, H.ptr(kw["yb"]), N, H.ptr(kw["stats_out"]), M, N, K, splitk, kw["relu"], None, 0, None, None,
                                 H.stream_ptr()), "bofi_rowgemm")
    torch.cuda.synchronize()
    scale = float(ref.abs().max())
    if kw["y"] is not None:
        y = kw["y"].cpu().double()
        y = y.sum(0) if mode == "splitk" else y
        assert float((y - ref).abs().max()) <= 2e-5 * max(1.0, scale) * (8 if mode == "ln_relu" else 1)
    if kw["yb"] is not None:
        assert float((kw["yb"].cpu().double() - ref).abs().max()) <= 1e-2 * max(1.0, scale)
    if kw["stats_out"] is not None:
        want = _row_stats(kw["y"].cpu(), N // 16)
        assert torch.allclose(kw["stats_out"].cpu(), want, rtol=1e-4, atol=1e-3)
    # the early-out word: nothing is written
    if kw["y"] is not None:
        kw["y"].fill_(7.0)
        skip = torch.tensor([5], dtype=torch.int32, device="cuda")
        H.check(H.lib().bofi_rowgemm(H.ptr(xc), K, H.ptr(wc), H.ptr(bc), H.ptr(kw["stats"]), kw["groups"], H.ptr(kw["colsum"]), H.ptr(kw["residual"]), N,
                                     H.ptr(kw["y"]), N, H.ptr(kw["yb"]), N, H.ptr(kw["stats_out"]), M, N, K, splitk, kw["relu"], H.ptr(skip), 5,
                                     None, None, H.stream_ptr()), "bofi_rowgemm")
        torch.cuda.synchronize()
        assert float((kw["y"] - 7.0).abs().max()) == 0.0


@pytest.mark.parametrize("B,R", [(1, 36), (7, 36), (64, 36), (13, 64), (256, 20)])
def test_bound_qattn(H, B, R):
    """bofi_bound_qattn (bound_ops.hip): folded-LayerNorm query projection + one query row of cross-attention per image and head,
    ragged region counts (one image without regions -> NaN, as softmax over an all-masked row), against float64."""
    g = _rng(11 + B)
    d, Hh, ld = 512, 8, 1024 + 512
    x32 = torch.randn(B, d, generator=g) * 1.2 + 0.1
    x = x32.to(torch.bfloat16)
    wq = (torch.randn(d, d, generator=g) / d ** 0.5).to(torch.bfloat16)
    bias = torch.randn(d, generator=g) * 0.1
    kv = (torch.randn(B * R, ld, generator=g)).to(torch.bfloat16)
    att_len = torch.randint(1, R + 1, (B,), generator=g).to(torch.int32)
    if B > 2:
        att_len[1] = 0
        att_len[2] = R
    mean = x32.double().mean(1, keepdim=True)
    rstd = 1.0 / (x32.double().std(1, keepdim=True) + 1e-6)
    q = rstd * (x.double() @ wq.double().t() - mean * wq.double().sum(1)) + bias.double()
    k = kv[:, :d].double().view(B, R, Hh, 64)
    v = kv[:, d:2 * d].double().view(B, R, Hh, 64)
    sc = torch.einsum("bhe,brhe->bhr", q.view(B, Hh, 64), k) / 8.0
    mask = torch.arange(R)[None, None, :] < att_len[:, None, None]
    p = torch.softmax(sc.masked_fill(~mask, float("-inf")), -1)
    ref = torch.einsum("bhr,brhe->bhe", p, v).reshape(B, d)
    out = torch.empty(B, d, dtype=torch.bfloat16, device="cuda")
    xc, wc, bc, cs, kvc, al = x.cuda(), wq.cuda(), bias.cuda(), wq.float().sum(1).cuda(), kv.cuda(), att_len.cuda()
    st = _row_stats(x32, 16).cuda()
    H.check(H.lib().bofi_bound_qattn(H.ptr(xc), H.ptr(st), H.ptr(wc), H.ptr(bc), H.ptr(cs), H.ptr(kvc), kvc.data_ptr() + d * 2, ld, H.ptr(al),
                                     H.ptr(out), B, R, None, 0, None, None, 0, H.stream_ptr()), "bofi_bound_qattn")
    torch.cuda.synchronize()
    got = out.cpu().double()
    empty = att_len == 0
    assert bool(got[empty].isnan().all()) and not bool(got[~empty].isnan().any())
    assert float((got[~empty] - ref[~empty]).abs().max()) <= 2e-2 * max(1.0, float(ref[~empty].abs().max()))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("n_rows", [0, 1, 100, 640])
def test_linear_over_a_row_list(H, dtype, n_rows):
    """bofi_linear_rows: only the listed rows are multiplied and written (count read on the device), the listed rows equal the dense
    bofi_linear's bit for bit, every other row of y keeps its old content."""
    g = _rng(21 + n_rows)
    M, N, K = 640, 512, 512
    code = H.DT_F32 if dtype == torch.float32 else H.DT_BF16
    x = torch.randn(M, K, generator=g).to(dtype).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dtype).cuda()
    bias = torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).cuda()
    dense = torch.empty(M, N, device="cuda")
    H.check(H.lib().bofi_linear(H.ptr(x), code, K, H.ptr(w), code, H.ptr(bias), H.ptr(res), N, H.ptr(dense), H.DT_F32, N, M, N, K, 1, None, 0, H.stream_ptr()))
    rows = torch.randperm(M, generator=g)[:n_rows].to(torch.int32)
    idx = torch.zeros(M, dtype=torch.int32)
    idx[:n_rows] = rows
    idx_d, cnt = idx.cuda(), torch.tensor([n_rows], dtype=torch.int32, device="cuda")
    y = torch.full((M, N), -3.0, device="cuda")
    H.check(H.lib().bofi_linear_rows(H.ptr(x), code, K, H.ptr(w), code, H.ptr(bias), H.ptr(res), N, H.ptr(y), H.DT_F32, N, M, N, K, 1, H.ptr(idx_d),
                                     H.ptr(cnt), H.stream_ptr()), "bofi_linear_rows")
    torch.cuda.synchronize()
    listed = torch.zeros(M, dtype=torch.bool)
    listed[rows.long()] = True
    assert torch.equal(y.cpu()[listed], dense.cpu()[listed])
    assert float((y.cpu()[~listed] + 3.0).abs().max()) == 0.0 if (~listed).any() else True


@pytest.mark.parametrize("n_rows", [0, 3, 70, 200])
def test_rowgemm_over_a_row_list_with_k_chunks(H, n_rows):
    """bofi_rowgemm with a device-side row list and K = 2048 walked in four chunks by one workgroup (no split-K): the listed rows get
    residual + x w^T + bias (float32 + bf16 copy + output statistics), every other row keeps its content."""
    g = _rng(31 + n_rows)
    M, N, K = 320, 512, 2048
    x = torch.randn(M, K, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16)
    bias, res = torch.randn(N, generator=g) * 0.1, torch.randn(M, N, generator=g)
    ref = x.double() @ w.double().t() + bias.double() + res.double()
    rows = torch.randperm(M, generator=g)[:n_rows].to(torch.int32)
    idx = torch.zeros(M, dtype=torch.int32); idx[:n_rows] = rows
    xc, wc, bc, idx_d, cnt = x.cuda(), w.cuda(), bias.cuda(), idx.cuda(), torch.tensor([n_rows], dtype=torch.int32, device="cuda")
    y = res.clone().cuda()                                     # the residual stream, updated in place at the listed rows
    yb = torch.full((M, N), -2.0, dtype=torch.bfloat16, device="cuda")
    st = torch.full((M, N // 16, 2), -1.0, device="cuda")
    H.check(H.lib().bofi_rowgemm(H.ptr(xc), K, H.ptr(wc), H.ptr(bc), None, 0, None, H.ptr(y), N, H.ptr(y), N, H.ptr(yb), N, H.ptr(st), M, N, K, 1, 0,
                                 None, 0, H.ptr(idx_d), H.ptr(cnt), H.stream_ptr()), "bofi_rowgemm")
    torch.cuda.synchronize()
    listed = torch.zeros(M, dtype=torch.bool); listed[rows.long()] = True
    got = y.cpu().double()
    if n_rows:
        assert float((got[listed] - ref[listed]).abs().max()) <= 3e-5 * max(1.0, float(ref.abs().max()))
        assert torch.allclose(st.cpu()[listed], _row_stats(y.cpu(), N // 16)[listed], rtol=1e-4, atol=1e-3)
        assert float((yb.cpu().double()[listed] - ref[listed]).abs().max()) <= 1e-2 * max(1.0, float(ref.abs().max()))
    assert torch.equal(y.cpu()[~listed], res[~listed]) and float((yb.cpu().float()[~listed] + 2.0).abs().max()) == 0.0
    assert float((st.cpu()[~listed] + 1.0).abs().max()) == 0.0


def test_bound_qattn_over_a_row_list(H):
    """bofi_bound_qattn with a device-side list of query rows, several rows per image (a decoder layer's cross-attention on the rows of
    the semi-autoregressive decode's new phrases): listed rows = the dense computation of those rows, others untouched."""
    g = _rng(77)
    Bimg, S, R, d, Hh, ld = 9, 20, 36, 512, 8, 1024
    Mrows = Bimg * S
    x32 = torch.randn(Mrows, d, generator=g) * 1.1
    x = x32.to(torch.bfloat16)
    wq = (torch.randn(d, d, generator=g) / d ** 0.5).to(torch.bfloat16)
    bias = torch.randn(d, generator=g) * 0.1
    kv = torch.randn(Bimg * R, ld, generator=g).to(torch.bfloat16)
    att_len = torch.randint(5, R + 1, (Bimg,), generator=g).to(torch.int32)
    mean = x32.double().mean(1, keepdim=True); rstd = 1.0 / (x32.double().std(1, keepdim=True) + 1e-6)
    q = (rstd * (x.double() @ wq.double().t() - mean * wq.double().sum(1)) + bias.double()).view(Mrows, Hh, 64)
    img = torch.arange(Mrows) // S
    k = kv[:, :d].double().view(Bimg, R, Hh, 64)[img]; v = kv[:, d:2 * d].double().view(Bimg, R, Hh, 64)[img]
    sc = torch.einsum("mhe,mrhe->mhr", q, k) / 8.0
    mask = torch.arange(R)[None, None, :] < att_len[img][:, None, None]
    ref = torch.einsum("mhr,mrhe->mhe", torch.softmax(sc.masked_fill(~mask, float("-inf")), -1), v).reshape(Mrows, d)
    rows = torch.randperm(Mrows, generator=g)[:37].to(torch.int32)
    idx = torch.zeros(Mrows, dtype=torch.int32); idx[:37] = rows
    out = torch.full((Mrows, d), -4.0, dtype=torch.bfloat16, device="cuda")
    xc, wc, bc, cs, kvc, al, idx_d = x.cuda(), wq.cuda(), bias.cuda(), wq.float().sum(1).cuda(), kv.cuda(), att_len.cuda(), idx.cuda()
    cnt, st = torch.tensor([37], dtype=torch.int32, device="cuda"), _row_stats(x32, 16).cuda()
    H.check(H.lib().bofi_bound_qattn(H.ptr(xc), H.ptr(st), H.ptr(wc), H.ptr(bc), H.ptr(cs), H.ptr(kvc), kvc.data_ptr() + d * 2, ld, H.ptr(al),
                                     H.ptr(out), Mrows, R, None, 0, H.ptr(idx_d), H.ptr(cnt), S, H.stream_ptr()), "bofi_bound_qattn")
    torch.cuda.synchronize()
    listed = torch.zeros(Mrows, dtype=torch.bool); listed[rows.long()] = True
    got = out.cpu().double()
    assert float((got[listed] - ref[listed]).abs().max()) <= 2e-2 * max(1.0, float(ref.abs().max()))
    assert float((got[~listed] + 4.0).abs().max()) == 0.0


@pytest.mark.parametrize("M,N,K", [(9216, 2048, 512), (11520 - 37, 1536, 512), (6600, 512, 2048), (2304, 6144, 512), (5760 - 37, 512, 512), (5760, 512, 2048)])
@pytest.mark.parametrize("mode", ["plain", "ln_relu", "res_stats"])
def test_linear_fused_persistent_equals_one_tile_per_workgroup(H, M, N, K, mode, monkeypatch):
    """bofi_linear_fused: the persistent 256 x 128 kernel (gemm_pers.hip: loader wavefronts, slab stream across tiles) against the
    one-tile-per-workgroup kernel (gemm_glds.hip) on the same operands -- every output BIT-equal (same MFMA K order, same epilogue
    order) -- and against float64 on the bf16-rounded operands.  Shapes: full tiles, a ragged last row tile, long K, wide N, and two of
    at most 128 tiles of 256 rows, which run on the 128-row form of the kernel (ragged, long K)."""
    g = _rng(M + N + K)
    x32 = torch.randn(M, K, generator=g) * 1.5 + 0.2
    x = x32.to(torch.bfloat16).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).cuda()
    bias = (torch.randn(N, generator=g) * 0.1).cuda()
    acc = None

    def run(pers):
        monkeypatch.setenv("BOFI_GEMM_PERS", "1" if pers else "0")
        H.lib().bofi_reload_env()
        kw = dict(res=None, y=None, y2=None, stats=None, colsum=None, groups=0, stats_out=None, relu=0, ydt=H.DT_F32)
        if mode == "plain":
            kw.update(y=torch.empty(M, N, dtype=torch.bfloat16, device="cuda"), ydt=H.DT_BF16)
        elif mode == "ln_relu":
            kw.update(stats=_row_stats(x32, K // 32).cuda(), colsum=w.double().sum(1).float(), relu=1,
                      y=torch.empty(M, N, dtype=torch.bfloat16, device="cuda"), ydt=H.DT_BF16)
        else:
            res = torch.randn(M, N, generator=_rng(5)).cuda()
            kw.update(res=res, y=res, y2=torch.empty(M, N, dtype=torch.bfloat16, device="cuda"), stats_out=torch.zeros(M, N // 32, 2, device="cuda"))     # in place, as the residual stream is
        H.check(H.lib().bofi_linear_fused(H.ptr(x), K, H.ptr(w), H.ptr(bias), H.ptr(kw["res"]), N, H.ptr(kw["y"]), kw["ydt"], N, H.ptr(kw["y2"]), N,
                                          H.ptr(kw["stats"]), H.ptr(kw["colsum"]), kw["groups"], H.ptr(kw["stats_out"]), M, N, K, kw["relu"], H.stream_ptr()),
                "bofi_linear_fused")
        torch.cuda.synchronize()
        return kw

    a, b = run(True), run(False)
    monkeypatch.delenv("BOFI_GEMM_PERS")
    H.lib().bofi_reload_env()
    for k in ("y", "y2", "stats_out"):
        if a[k] is not None:
            assert torch.equal(a[k], b[k]), k
    acc = (x.double() @ w.double().t()).cpu()
    if mode == "plain":
        ref = acc + bias.cpu().double()
    elif mode == "ln_relu":
        mean = x32.double().mean(1, keepdim=True)
        rstd = 1.0 / (x32.double().std(1, keepdim=True) + 1e-6)
        ref = torch.relu(rstd * (acc - mean * w.double().sum(1).cpu()) + bias.cpu().double())
    else:
        ref = acc + bias.cpu().double() + torch.randn(M, N, generator=_rng(5)).double()
    scale = max(1.0, float(ref.abs().max()))
    y = a["y"].cpu().double()
    assert float((y - ref).abs().max()) <= (1e-2 if a["y"].dtype == torch.bfloat16 else 2e-5 * 8) * scale
    if a["stats_out"] is not None:
        assert torch.allclose(a["stats_out"].cpu(), _row_stats(a["y"].cpu(), N // 32), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("M,N,K", [(300, 128, 192), (257, 256, 320), (64, 1024, 512), (1000, 384, 512)])
@pytest.mark.parametrize("mode", ["plain", "ln", "res"])
def test_linear_fused_persistent_small_and_ragged_shapes(H, M, N, K, mode, monkeypatch):
    """The persistent kernel forced onto shapes it is not picked for (BOFI_GEMM_PERS_MIN=1): one or two row tiles, a single column tile,
    the shortest K it takes (3 slabs), fewer rows than a tile -- bit-equal to the one-tile kernel.  (Folded LayerNorm needs 16 statistic
    groups: K = 512 only; the other K values fall back to the one-tile kernel on both sides.)"""
    g = _rng(7 * M + N + K)
    x32 = torch.randn(M, K, generator=g) * 1.5 + 0.2
    x = x32.to(torch.bfloat16).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).cuda()
    bias = (torch.randn(N, generator=g) * 0.1).cuda()
    outs = []
    for pers in ("1", "0"):
        monkeypatch.setenv("BOFI_GEMM_PERS", pers); monkeypatch.setenv("BOFI_GEMM_PERS_MIN", "1")
        H.lib().bofi_reload_env()
        if mode == "res":
            y = torch.randn(M, N, generator=_rng(5)).cuda()
            y2 = torch.empty(M, N, dtype=torch.bfloat16, device="cuda"); so = torch.zeros(M, N // 32, 2, device="cuda")
            args = (H.ptr(y), N, H.ptr(y), H.DT_F32, N, H.ptr(y2), N, None, None, 0, H.ptr(so))
            keep = (y, y2, so)
        else:
            y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
            st = _row_stats(x32, K // 32).cuda() if mode == "ln" else None
            cs = w.double().sum(1).float() if mode == "ln" else None
            args = (None, N, H.ptr(y), H.DT_BF16, N, None, N, H.ptr(st), H.ptr(cs), 0, None)
            keep = (y,)
        H.check(H.lib().bofi_linear_fused(H.ptr(x), K, H.ptr(w), H.ptr(bias), *args, M, N, K, 1 if mode == "ln" else 0, H.stream_ptr()), "bofi_linear_fused")
        torch.cuda.synchronize()
        outs.append(keep)
    monkeypatch.delenv("BOFI_GEMM_PERS"); monkeypatch.delenv("BOFI_GEMM_PERS_MIN")
    H.lib().bofi_reload_env()
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    ref = (x.double() @ w.double().t()).cpu() + bias.cpu().double()
    if mode == "ln":
        mean = x32.double().mean(1, keepdim=True); rstd = 1.0 / (x32.double().std(1, keepdim=True) + 1e-6)
        ref = torch.relu(rstd * ((x.double() @ w.double().t()).cpu() - mean * w.double().sum(1).cpu()) + bias.cpu().double())
    elif mode == "res":
        ref = ref + torch.randn(M, N, generator=_rng(5)).double()
    assert float((outs[0][0].cpu().double() - ref).abs().max()) <= 2e-2 * max(1.0, float(ref.abs().max()))


def test_vocab_finalize_reads_a_padded_source(H):
    """bofi vocab_finalize with the logits in a buffer of another pitch (the bf16 engine's generator output, pitch 9 600): same ids and
    log-probs as in place, through the engine: BOFI_GEN_PAD=0 (in place, one-tile GEMM) against the default on one decode."""
    import os, subprocess, sys, json
    code = (
        "import os, sys, json, torch\n"
        "sys.path.insert(0, os.getcwd())\n"
        "from boficap_amd import weights as W\n"
        "from boficap_amd.config import FULL as cfg\n"
        "from boficap_amd.engine import BofiEngine\n"
        "sd = W.make_state_dict(cfg, seed=0, gen_scale=6.0)\n"
        "eng = BofiEngine(cfg, torch.bfloat16, max_batch=8, max_regions=36); eng.load_state_dict(sd)\n"
        "att = torch.from_numpy(W.synthetic_att_feats(8, 36, cfg.att_feat_size, seed=77)).cuda().to(torch.bfloat16)\n"
        "r = eng.decode_naic(att)\n"
        "torch.cuda.synchronize()\n"
        "seq, lp = r['seq'], r['seq_logprob']\n"
        "print(json.dumps({'seq': seq.cpu().tolist(), 'lp_sum': float(torch.nan_to_num(lp.float()).double().sum()), 'lp_max': float(torch.nan_to_num(lp.float()).max())}))\n")
    outs = []
    for pad in ("1", "0"):
        env = dict(os.environ, BOFI_GEN_PAD=pad)
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), timeout=300)
        assert p.returncode == 0, p.stderr[-2000:]
        outs.append(json.loads(p.stdout.strip().splitlines()[-1]))
    assert outs[0]["seq"] == outs[1]["seq"]
    assert abs(outs[0]["lp_sum"] - outs[1]["lp_sum"]) <= 1e-3 * max(1.0, abs(outs[1]["lp_sum"]))
