"""End-to-end NAIC bound+fill parity on the MI355X through the C ABI / the drop-in module."""
import os

import numpy as np
import pytest
import torch

import boficap_oracle as O
from conftest import TINY_CASES, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engines(weight_cache, manifest):
    from boficap_amd.engine import BofiEngine
    cache = {}

    def get(case, dtype, max_batch=64):
        m = manifest[case]
        key = (m["config"], m["seed"], m["gen_scale"], m.get("patch"), dtype, max_batch)
        if key not in cache:
            cfg, sd = weight_cache(m["config"], m["seed"], m["gen_scale"], m["digest"], m.get("patch"))
            e = BofiEngine(cfg, dtype, max_batch=max_batch, max_regions=36)
            e.load_state_dict(sd)
            cache[key] = (cfg, sd, e)
        return cache[key]
    return get


def _inputs(g, device="cuda"):
    att = torch.from_numpy(g["att_feats"]).to(device)
    att_len = torch.from_numpy(g["att_masks"]).sum(1).to(torch.int32).to(device) if "att_masks" in g else None
    return att, att_len


def _close(a, b, tol):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    assert a.shape == b.shape
    assert (np.isnan(a) == np.isnan(b)).all(), "NaN pattern differs"
    m = ~np.isnan(a) & ~np.isinf(a)
    return float(np.abs(a[m] - b[m]).max()) if m.any() else 0.0


@pytest.mark.parametrize("name", TINY_CASES)
def test_golden_tiny_f32(name, engines):
    """fp32 engine vs vectors recorded from the reference: ids and slots bit-exact, logits <= 1e-3."""
    cfg, sd, eng = engines(name, torch.float32)
    g = load_golden(name)
    att, att_len = _inputs(g)
    mem = eng.encode(att, att_len)
    R = g["memory"].shape[1]                            # the reference clips to the longest image
    assert _close(mem.cpu().numpy()[:, :R], g["memory"], 0) < 1e-4
    B = att.size(0)
    ext = torch.zeros(B, cfg.bound_len, dtype=torch.int32, device="cuda"); ext[:, 0] = cfg.len_idx
    last = torch.ones(B, dtype=torch.int32, device="cuda")
    llp, slp = eng.bound_step(ext, last, att.size(1), att_len)
    assert _close(llp.cpu().numpy(), g["step0_len_logp"], 0) < 1e-4
    assert _close(slp.cpu().numpy(), g["step0_syn_logp"], 0) < 1e-4
    for graph in (False, True, True):
        r = eng.decode_naic(att, att_len, graph=graph, out=None if not graph else r)
        torch.cuda.synchronize()
        assert (r["phrase_num"].cpu().numpy() == g["naic_phrase_num"]).all()
        assert (r["phrase_length"].cpu().numpy() == g["naic_phrase_length"]).all()
        assert (r["phrase_syn"].cpu().numpy() == g["naic_phrase_syn"]).all()
        assert (r["seq"].cpu().numpy() == g["naic_seq"]).all()
        assert int(r["bound_iters"]) == int(g["naic_iters"])
        assert _close(r["seq_logprob"].cpu().numpy(), g["naic_logprob"], 0) < 1e-3
        assert r["seq"].dtype == torch.int64 and r["phrase_num"].dtype == torch.int32
        assert r["phrase_length"].dtype == torch.int32 and r["phrase_syn"].dtype == torch.int64


@pytest.mark.parametrize("name", TINY_CASES + ["full_b8"])
def test_golden_saic_f32(name, engines, manifest):
    """Semi-autoregressive decode (core_SAIC) vs vectors recorded from the reference, including the early
    return on NaN (an image whose first phrase is empty)."""
    from boficap_amd import weights as W
    cfg, sd, eng = engines(name, torch.float32)
    g = load_golden(name)
    if name == "full_b8":
        m = manifest[name]
        att = torch.from_numpy(W.synthetic_att_feats(m["pool_size"], 36, cfg.att_feat_size, seed=m["pool_seed"])[g["pool_index"]]).cuda()
        att_len = None
    else:
        att, att_len = _inputs(g)
    r = eng.decode_saic(att, att_len)
    torch.cuda.synchronize()
    assert (r["phrase_num"].cpu().numpy() == g["saic_phrase_num"]).all()
    assert (r["phrase_length"].cpu().numpy() == g["saic_phrase_length"]).all()
    assert (r["phrase_syn"].cpu().numpy() == g["saic_phrase_syn"]).all()
    assert (r["seq"].cpu().numpy() == g["saic_seq"]).all()
    lp = r["seq_logprob"].cpu()
    if "saic_logprob" in g:
        assert _close(lp.numpy(), g["saic_logprob"], 0) < 1e-3
    else:
        assert _close(torch.topk(lp, 2, dim=2)[0].numpy(), g["saic_top2_val"], 0) < 1e-3


def test_golden_full_f32(engines, manifest):
    from boficap_amd import weights as W
    cfg, sd, eng = engines("full_b8", torch.float32)
    m, g = manifest["full_b8"], load_golden("full_b8")
    att = torch.from_numpy(W.synthetic_att_feats(m["pool_size"], 36, cfg.att_feat_size, seed=m["pool_seed"])[g["pool_index"]]).cuda()
    r = eng.decode_naic(att, want_memory=True)
    torch.cuda.synchronize()
    assert _close(r["memory"].cpu().numpy()[:2], g["memory"], 0) < 1e-4
    assert (r["phrase_num"].cpu().numpy() == g["naic_phrase_num"]).all()
    assert (r["phrase_length"].cpu().numpy() == g["naic_phrase_length"]).all()
    assert (r["phrase_syn"].cpu().numpy() == g["naic_phrase_syn"]).all()
    assert (r["seq"].cpu().numpy() == g["naic_seq"]).all()
    lp = r["seq_logprob"].cpu()
    assert _close(lp[:2, :3].numpy(), g["naic_logprob_rows"], 0) < 1e-3
    top = torch.topk(lp, 2, dim=2)
    assert _close(top[0].numpy(), g["naic_top2_val"], 0) < 1e-3


def test_bf16_engine_takes_float32_or_bf16_features(weight_cache):
    """A bf16 engine converts float32 features once (engine.hip: launch_cast_bf16): the decode equals the one on features the caller
    rounded.  (The bf16 tolerance itself is shown on EVERY image, teacher-forced, in test_gpu_full.py::
    test_bf16_logits_within_tolerance_on_every_image -- this test used to bound it on the 70 % of images whose layouts survived.)"""
    from boficap_amd import weights as W
    from boficap_amd.engine import BofiEngine
    from boficap_amd.config import FULL as cfg
    sd = weight_cache("FULL", 0, 1.0)[1]
    att_np = W.synthetic_att_feats(32, 36, cfg.att_feat_size, seed=99)
    eng = BofiEngine(cfg, torch.bfloat16, max_batch=32, max_regions=36)
    eng.load_state_dict(sd)
    a = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in eng.decode_naic(torch.from_numpy(att_np).cuda(), strict_q1=False).items()}
    b = eng.decode_naic(torch.from_numpy(att_np).cuda().to(torch.bfloat16), strict_q1=False)
    torch.cuda.synchronize()
    for k in ("seq", "phrase_num", "phrase_length", "phrase_syn"):
        assert torch.equal(a[k], b[k]), k
    assert torch.equal(a["seq_logprob"].isnan(), b["seq_logprob"].isnan())
    assert float((a["seq_logprob"] - b["seq_logprob"]).nan_to_num().abs().max()) == 0.0


def test_full_batch_properties(engines, weight_cache):
    """B = 64 at the benchmark shape: oracle parity on the slots, and size-independent properties:
    idempotence, graph == eager, per-image determinism of the bounding pass (Q2) and, with the
    Q1 fix, batch-composition independence of the ids."""
    from boficap_amd import weights as W
    cfg, sd, eng = engines("full_b8", torch.float32)
    att_np = W.synthetic_att_feats(64, 36, cfg.att_feat_size, seed=77)
    att = torch.from_numpy(att_np).cuda()
    a = eng.decode_naic(att)
    b = eng.decode_naic(att, graph=True)
    c = eng.decode_naic(att, graph=True, out=b)
    torch.cuda.synchronize()
    for k in ("seq", "phrase_num", "phrase_length", "phrase_syn"):
        assert torch.equal(a[k], b[k]) and torch.equal(a[k], c[k])
    assert torch.equal(a["seq_logprob"].isnan(), b["seq_logprob"].isnan())
    assert (a["seq_logprob"] - b["seq_logprob"]).nan_to_num().abs().max() == 0
    # slots are a per-image function: any sub-batch gives the same rows
    sub = eng.decode_naic(att[16:48].contiguous())
    for k in ("phrase_num", "phrase_length", "phrase_syn"):
        assert torch.equal(sub[k], a[k][16:48])
    # with the per-row fill mask the ids are per-image too
    f_all = eng.decode_naic(att, strict_q1=False)
    f_sub = eng.decode_naic(att[16:48].contiguous(), strict_q1=False)
    assert torch.equal(f_sub["seq"], f_all["seq"][16:48])
    # oracle on the same batch (CPU, a few seconds): slot layout bit-exact; ids where the gap allows
    w = O.as_torch(sd)
    seq, lp, pn, pl, ps, _ = O.sample_naic(w, cfg, torch.from_numpy(att_np))
    assert torch.equal(a["phrase_num"].cpu(), pn) and torch.equal(a["phrase_length"].cpu(), pl) and torch.equal(a["phrase_syn"].cpu(), ps)
    if not lp.isnan().any():
        top = torch.topk(lp, 2, dim=2)[0]
        safe = (top[..., 0] - top[..., 1]) > 1e-3
        assert torch.equal(a["seq"].cpu()[safe], seq[safe])
        assert (a["seq_logprob"].cpu() - lp).abs().max() < 1e-3


@pytest.mark.parametrize("name", ["tiny_mix", "full_b8"])
def test_iterative_refine_vs_oracle(name, engines, manifest):
    """BASELINE config 5: refinement rounds of the filling pass.  No reference implementation exists; the oracle
    defines it on the reference's glat_input hook and the engine must match the oracle."""
    from boficap_amd import weights as W
    cfg, sd, eng = engines(name, torch.float32)
    g = load_golden(name)
    if name == "full_b8":
        m = manifest[name]
        att = torch.from_numpy(W.synthetic_att_feats(m["pool_size"], 36, cfg.att_feat_size, seed=m["pool_seed"])[g["pool_index"]])
    else:
        att = torch.from_numpy(g["att_feats"])
    w = O.as_torch(sd)
    for rounds in (1, 3):
        oseq, olp, opn, opl, ops = O.sample_naic_refine(w, cfg, att, rounds=rounds)
        r = eng.decode_naic(att.cuda(), refine_rounds=rounds)
        torch.cuda.synchronize()
        assert torch.equal(r["phrase_length"].cpu(), opl)
        assert float((r["seq_logprob"].cpu() - olp).nan_to_num().abs().max()) < 1e-3
        top = torch.topk(olp, 2, dim=2)[0]
        safe = (top[..., 0] - top[..., 1]) > 1e-3
        assert torch.equal(r["seq"].cpu()[safe], oseq[safe])


def test_forked_engines_in_flight(engines):
    """Several decodes in flight on separate streams (engine forks sharing the weights) give exactly
    the results of one-at-a-time decodes, for equal and for different inputs."""
    from boficap_amd import weights as W
    cfg, sd, eng = engines("full_b8", torch.bfloat16)
    atts = [torch.from_numpy(W.synthetic_att_feats(64, 36, cfg.att_feat_size, seed=500 + i)).cuda().to(torch.bfloat16) for i in range(3)]
    ref = [eng.decode_naic(a) for a in atts]
    torch.cuda.synchronize()
    forks = [eng, eng.fork(), eng.fork()]
    streams = [torch.cuda.Stream() for _ in forks]
    outs = [None] * 3
    for rep in range(4):                                   # first round captures the graphs, later ones replay
        for k, (e, st) in enumerate(zip(forks, streams)):
            with torch.cuda.stream(st):
                outs[k] = e.decode_naic(atts[k], graph=True, out=outs[k])
    torch.cuda.synchronize()
    for r, o in zip(ref, outs):
        for key in ("seq", "phrase_num", "phrase_length", "phrase_syn", "bound_iters"):
            assert torch.equal(r[key], o[key]), key
        assert torch.equal(r["seq_logprob"].isnan(), o["seq_logprob"].isnan())
        assert (r["seq_logprob"] - o["seq_logprob"]).nan_to_num().abs().max() == 0
    del forks, outs


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_ragged_regions_full_config(dtype, weight_cache):
    """Full-size model, 50 regions with per-image region counts (att_masks), batch sizes 1 and 5: the padded
    rows must be exact zeros after att_embed and masked as keys everywhere (AttModel.py:46-51, 113-120)."""
    from boficap_amd import weights as W
    from boficap_amd.engine import BofiEngine
    cfg, sd = weight_cache("FULL", 0, 1.0)
    w = O.as_torch(sd)
    eng = BofiEngine(cfg, dtype, max_batch=8, max_regions=50)
    eng.load_state_dict(sd)
    att_np = W.synthetic_att_feats(5, 50, cfg.att_feat_size, seed=21)
    lens = [50, 17, 36, 1, 44]
    masks = np.zeros((5, 50), np.float32)
    for b, n in enumerate(lens):
        masks[b, :n] = 1
        att_np[b, n:] = 7.0                                  # garbage in the padding must not leak
    for sl in (slice(0, 5), slice(1, 2)):
        att, am = torch.from_numpy(att_np[sl]), torch.from_numpy(masks[sl])
        oseq, olp, opn, opl, ops, _ = O.sample_naic(w, cfg, att, am, fix_q1=True)
        r = eng.decode_naic(att.cuda(), am.sum(1).to(torch.int32).cuda(), strict_q1=False)
        torch.cuda.synchronize()
        if dtype == torch.float32:
            assert torch.equal(r["phrase_length"].cpu(), opl) and torch.equal(r["phrase_syn"].cpu(), ops)
            assert float((r["seq_logprob"].cpu() - olp).nan_to_num().abs().max()) < 1e-3
        else:
            same = (r["phrase_length"].cpu() == opl).all(1) & (r["phrase_syn"].cpu() == ops).all(1)      # (a flipped label changes the fill input)
            assert int(same.sum()) >= len(same) - 2                # (free decode: ~10 % of the layouts flip on a near-tie in bf16, test_gpu_full's every-image test)
            assert float((r["seq_logprob"].cpu()[same] - olp[same]).nan_to_num().abs().max()) < 2e-2


def test_engine_rejects_bad_calls(engines):
    from boficap_amd.hip import BofiHipError
    cfg, sd, eng = engines("tiny_mix", torch.float32)
    good = torch.zeros(2, 36, cfg.att_feat_size, device="cuda")
    with pytest.raises(BofiHipError):
        eng.decode_naic(torch.zeros(2, 36, cfg.att_feat_size + 64, device="cuda"))      # wrong feature size
    with pytest.raises(BofiHipError):
        eng.decode_naic(torch.zeros(65, 36, cfg.att_feat_size, device="cuda"))          # batch > max_batch
    with pytest.raises(BofiHipError):
        eng.decode_naic(torch.zeros(2, 40, cfg.att_feat_size, device="cuda"))           # regions > max_regions
    with pytest.raises(BofiHipError):
        eng.decode_naic(good.cpu())                                                      # host tensor
    with pytest.raises(BofiHipError):
        eng.decode_naic(good.to(torch.bfloat16))                                         # bf16 features into an f32 engine
    with pytest.raises(BofiHipError):
        eng.decode_naic(good, torch.ones(2, dtype=torch.int64, device="cuda"))           # att_len must be int32
    r = eng.decode_naic(good)                                                            # and the engine still works afterwards
    torch.cuda.synchronize()
    assert r["seq"].shape == (2, cfg.seq_length)


def test_drop_in_module_sample(weight_cache, manifest):
    """captioning.models.setup(opt) -> load_state_dict -> model(..., mode='sample'): the 6-tuple."""
    import captioning.models as models
    m = manifest["tiny_mix"]
    cfg, sd = weight_cache(m["config"], m["seed"], m["gen_scale"], m["digest"], m.get("patch"))
    g = load_golden("tiny_mix")
    model = models.setup(cfg.to_opt())
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model.cuda().eval()
    att = torch.from_numpy(g["att_feats"]).cuda()
    fc = torch.zeros(att.size(0), 0, device="cuda")
    with torch.no_grad():
        seq, lp, pn, pl, ps, secs = model(fc, att, None, opt={"train_mode": "NAIC", "sample_method": "greedy", "sample_n": 1}, mode="sample")
        _, _, memory, masks = model._prepare_feature(fc, att, None)
    assert (seq.cpu().numpy() == g["naic_seq"]).all() and (pl.cpu().numpy() == g["naic_phrase_length"]).all()
    assert (pn.cpu().numpy() == g["naic_phrase_num"]).all() and (ps.cpu().numpy() == g["naic_phrase_syn"]).all()
    assert _close(lp.cpu().numpy(), g["naic_logprob"], 0) < 1e-3
    assert isinstance(secs, float) and masks.shape == (att.size(0), 1, 36)
    assert _close(memory.cpu().numpy(), g["memory"], 0) < 1e-4
    seq, lp, pn, pl, ps, secs = model(fc, att, None, opt={"train_mode": "SAIC", "sample_method": "greedy"}, mode="sample")
    assert (seq.cpu().numpy() == g["saic_seq"]).all() and (pl.cpu().numpy() == g["saic_phrase_length"]).all()
    with pytest.raises(NotImplementedError):
        model(fc, att, None, opt={"train_mode": "AIC"}, mode="sample")


def _integration_stub_b():
    """The fenced python block of INTEGRATION.md section B (the ctypes stub a maintainer of the reference would paste into AttModel.py)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "INTEGRATION.md")) as f:
        text = f.read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    stub = [b for b in blocks if "replacement for the 'NAIC' branch of _sample" in b]
    assert len(stub) == 1, "INTEGRATION.md: stub B not found"
    return stub[0], root


def test_integration_stub_b(weight_cache, manifest):
    """the binding INTEGRATION.md documents is EXECUTED -- the block is extracted from the file, pointed at the built library, given an object that
    carries the attributes the reference's model has (AttModel.py:56-79, TransformerModel.py:1631-1640) and its state_dict, and compared with model(..., mode='sample') and the
    REFERENCE's full_b8 fixture: float32 engine, ids and slot layout bit-exact, log-probs <= 1e-3 (AttModel.py:419-429)."""
    import types
    import captioning.models as models
    from boficap_amd import hip, weights as W
    src, root = _integration_stub_b()
    assert "bofi_abi_version() == %d" % hip.ABI_VERSION in src, "stub B asserts another ABI version than the library's"
    lib_path = os.path.join(root, "boficap_amd", "libboficap_hip.so")
    src = src.replace('C.CDLL("libboficap_hip.so")', "C.CDLL(%r)" % lib_path)
    ns = {}
    exec(compile(src, "INTEGRATION.md:stub_b", "exec"), ns)
    ns["_DTYPE"] = 0                                             # the bit-exact parity engine
    m, g = manifest["full_b8"], load_golden("full_b8")
    cfg, sd = weight_cache(m["config"], m["seed"], m["gen_scale"], m["digest"], m.get("patch"))
    model = models.setup(cfg.to_opt())
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model.cuda().eval()
    # what the pasted functions read from `self` in the reference's class
    ref_self = types.SimpleNamespace(tgt_vocab=cfg.tgt_vocab, att_feat_size=cfg.att_feat_size, d_model=cfg.d_model, d_ff=cfg.d_ff, h=cfg.h, N_enc=cfg.N_enc,
                                     N_dec=cfg.N_dec, N_len=cfg.N_len, seq_length=cfg.seq_length, state_dict=model.state_dict)
    ref_self._bofi = ns["_engine"](ref_self)
    att = torch.from_numpy(W.synthetic_att_feats(m["pool_size"], 36, cfg.att_feat_size, seed=m["pool_seed"])[g["pool_index"]]).cuda()
    fc = torch.zeros(att.size(0), 0, device="cuda")
    seq, logp, pn, pl, ps = ns["_sample_naic"](ref_self, att, None)
    torch.cuda.synchronize()
    with torch.no_grad():
        mseq, mlp, mpn, mpl, mps, _ = model(fc, att, None, opt={"train_mode": "NAIC", "sample_method": "greedy", "sample_n": 1}, mode="sample")
    for a, b, name in ((seq, mseq, "seq"), (pn, mpn, "phrase_num"), (pl, mpl, "phrase_length"), (ps, mps, "phrase_syn")):
        assert a.dtype == b.dtype and torch.equal(a, b), name
    assert _close(logp.cpu().numpy(), mlp.cpu().numpy(), 0) < 1e-3
    # ... and the reference's own outputs for these images
    assert (seq.cpu().numpy() == g["naic_seq"]).all() and (pn.cpu().numpy() == g["naic_phrase_num"]).all()
    assert (pl.cpu().numpy() == g["naic_phrase_length"]).all() and (ps.cpu().numpy() == g["naic_phrase_syn"]).all()
    assert _close(torch.topk(logp.cpu(), 2, dim=2)[0].numpy(), g["naic_top2_val"], 0) < 1e-3
    # ragged regions through the stub's att_masks argument: the model's own answer on the same call
    masks = torch.ones(att.size(0), 36, device="cuda"); masks[1, 20:] = 0; masks[3, 30:] = 0
    seq2, logp2, pn2, pl2, ps2 = ns["_sample_naic"](ref_self, att, masks)
    with torch.no_grad():
        mseq2, mlp2, mpn2, mpl2, _, _ = model(fc, att, masks, opt={"train_mode": "NAIC", "sample_method": "greedy"}, mode="sample")
    assert torch.equal(seq2, mseq2) and torch.equal(pn2, mpn2) and torch.equal(pl2, mpl2)
    ns["_lib"].bofi_engine_destroy(ref_self._bofi)


def test_bounding_loop_with_fewer_iterations_enqueued(weight_cache, manifest):
    """bofi_engine_set_bound_iter_cap through model(..., mode='sample'): the loop enqueued for c iterations, the count of live iterations
    read back, the decode repeated without the cap when c was not enough -- the reference's 6-tuple whatever c (opt.bofi_naic_iter_cap),
    and the adaptive choice (recent decodes + 2) after three calls."""
    import captioning.models as models
    m = manifest["tiny_mix"]
    cfg, sd = weight_cache(m["config"], m["seed"], m["gen_scale"], m["digest"], m.get("patch"))
    g = load_golden("tiny_mix")
    model = models.setup(cfg.to_opt())
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model.cuda().eval()
    att = torch.from_numpy(g["att_feats"]).cuda()
    fc = torch.zeros(att.size(0), 0, device="cuda")
    eng = model.engine()
    whole = eng.decode_naic(model._as_input(att), None)
    live = int(whole["bound_iters"])
    assert 1 <= live < cfg.seq_length
    short = eng.decode_naic(model._as_input(att), None, iter_cap=1)
    assert int(short["bound_iters"]) == 1                              # "may not be through": the caller decodes again
    enough = eng.decode_naic(model._as_input(att), None, iter_cap=live + 1)
    assert int(enough["bound_iters"]) == live and torch.equal(enough["seq"], whole["seq"]) and torch.equal(enough["phrase_length"], whole["phrase_length"])
    for cap in (None, 0, 1, live, live + 1, cfg.seq_length, None, None, None):
        model.opt.bofi_naic_iter_cap = cap
        with torch.no_grad():
            seq, lp, pn, pl, ps, _ = model(fc, att, None, opt={"train_mode": "NAIC", "sample_method": "greedy", "sample_n": 1}, mode="sample")
        assert (seq.cpu().numpy() == g["naic_seq"]).all() and (pl.cpu().numpy() == g["naic_phrase_length"]).all(), cap
        assert (pn.cpu().numpy() == g["naic_phrase_num"]).all() and (ps.cpu().numpy() == g["naic_phrase_syn"]).all(), cap
        assert _close(lp.cpu().numpy(), g["naic_logprob"], 0) < 1e-3
    assert model._naic_recent[-1] == live and eng._iter_cap == min(cfg.seq_length, live + 2) % cfg.seq_length      # the last calls ran under the adaptive cap


def test_fork_right_after_a_capped_decode_does_not_inherit_the_cap(engines):
    """bofi_engine_fork copies the parent's struct: the per-call knobs (bounding-iteration cap, live-iteration word, semi-autoregressive
    range) must start from their defaults in the fork, or a fork made behind a capped decode (sample_pair in the self-critical step behind a
    periodic eval) runs a truncated bounding loop that nobody checks."""
    cfg, sd, eng = engines("tiny_mix", torch.float32)
    att, att_len = _inputs(load_golden("tiny_mix"))
    whole = eng.decode_naic(att, att_len)
    live = int(whole["bound_iters"])
    assert live >= 2
    word = torch.zeros(1, dtype=torch.int32, device="cuda")
    eng.watch_live_iterations(word)
    short = eng.decode_naic(att, att_len, iter_cap=1)                  # leaves cap = 1 and the word set on the parent
    assert int(short["bound_iters"]) == 1 and int(word) == 1
    f = eng.fork()
    assert (f._iter_cap, f._q1_group, f._live_word) == (0, 0, None)
    word.zero_()
    r = f.decode_naic(att, att_len)                                    # iter_cap = 0 == the handle's default: no call resets the native cap
    assert int(r["bound_iters"]) == live and int(word) == 0           # the whole loop, and the parent's word left alone
    for k in ("seq", "phrase_num", "phrase_length", "phrase_syn"):
        assert torch.equal(r[k], whole[k]), k
    eng.watch_live_iterations(None)
    eng.decode_naic(att, att_len)                                      # (parent back to its defaults for the tests that share it)


def test_entropy_perplexity_without_materialising_logprobs(engines):
    """eval's per-image entropy / perplexity (eval_utils.py:463-464) from the fused row reductions, with the log-prob
    tensor in user memory and with it left in the engine's workspace."""
    import torch.nn.functional as F
    cfg, sd, eng = engines("tiny_mix", torch.float32)
    att, att_len = _inputs(load_golden("tiny_mix"))
    r = eng.decode_naic(att, att_len)
    lp, seq = r["seq_logprob"], r["seq"]
    denom = (seq > 0).to(lp).sum(1) + 1
    ent_ref = -(F.softmax(lp, dim=2) * lp).sum(2).sum(1) / denom
    ppl_ref = -lp.gather(2, seq.unsqueeze(2)).squeeze(2).sum(1) / denom
    ent, ppl = eng.entropy_perplexity(r)
    assert _close(ent.cpu().numpy(), ent_ref.cpu().numpy(), 0) < 1e-4 and _close(ppl.cpu().numpy(), ppl_ref.cpu().numpy(), 0) < 1e-4
    r2 = eng.decode_naic(att, att_len, want_logprob=False)
    assert r2["seq_logprob"] is None and torch.equal(r2["seq"], seq)
    ent2, ppl2 = eng.entropy_perplexity(r2)
    assert _close(ent2.cpu().numpy(), ent_ref.cpu().numpy(), 0) < 1e-4 and _close(ppl2.cpu().numpy(), ppl_ref.cpu().numpy(), 0) < 1e-4
    # ... and straight out of the decode's vocabulary epilogue (bofi_engine_set_row_stats_out: no second pass over the tensor), eager and replayed, with refinement
    # rounds, with the log-probs materialised or not; the per-position figures against the read-back kernel's
    for kw in (dict(), dict(want_logprob=False), dict(graph=True), dict(refine_rounds=1)):
        plain = eng.decode_naic(att, att_len, **kw)
        p_ref, c_ref = eng.row_stats(plain)
        p_ref, c_ref = p_ref.clone(), c_ref.clone()
        r3 = eng.decode_naic(att, att_len, row_stats=True, **kw)
        if kw.get("graph"):
            r3 = eng.decode_naic(att, att_len, row_stats=True, out=r3, **kw)
        torch.cuda.synchronize()
        assert torch.equal(r3["seq"], plain["seq"]) and r3["_row_stats_fused"]
        p3, c3 = eng.row_stats(r3)
        assert p3.data_ptr() == r3["row_plogp"].data_ptr()
        assert float((p3 - p_ref).abs().max()) < 2e-5 and torch.equal(c3, c_ref)
        if not kw:
            ent3, ppl3 = eng.entropy_perplexity(r3)
            assert _close(ent3.cpu().numpy(), ent_ref.cpu().numpy(), 0) < 1e-4 and _close(ppl3.cpu().numpy(), ppl_ref.cpu().numpy(), 0) < 1e-4
    r4 = eng.decode_naic(att, att_len)                          # (off again: the next plain decode leaves the buffers alone)
    assert not r4["_row_stats_fused"]


def test_sampled_tokens_follow_the_fill_distribution(weight_cache, manifest):
    """sample_method='sample' (Categorical(logits = logp / T), CaptionModel.py:405-425): empirical frequencies of many
    draws against softmax(logp / T), pad after the caption length, the reference's row layout for sample_n."""
    import captioning.models as models
    m = manifest["tiny_mix"]
    cfg, sd = weight_cache(m["config"], m["seed"], m["gen_scale"], m["digest"], m.get("patch"))
    g = load_golden("tiny_mix")
    model = models.setup(cfg.to_opt())
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model.cuda().eval()
    att = torch.from_numpy(g["att_feats"]).cuda()
    fc = torch.zeros(att.size(0), 0, device="cuda")
    B, n, T = att.size(0), 3000, 0.7
    with torch.no_grad():
        seq, lp, pn, pl, ps, _ = model(fc, att, None, opt={"train_mode": "NAIC", "sample_method": "sample", "sample_n": n, "temperature": T},
                                       mode="sample")
        greedy = model(fc, att, None, opt={"train_mode": "NAIC", "sample_method": "greedy"}, mode="sample")
    assert seq.shape == (B * n, cfg.seq_length) and lp.shape[0] == B * n and pl.shape[0] == B * n
    assert torch.equal(pl[::n], greedy[3]) and torch.equal(lp[::n], greedy[1])              # layouts / distributions are per image
    ntok = greedy[3].sum(1)
    seq = seq.view(B, n, -1)
    for b in range(B):
        k = int(ntok[b])
        assert (seq[b, :, k:] == 0).all()
        if k == 0 or greedy[1][b].isnan().any():
            continue
        probs = torch.softmax(greedy[1][b, 0] / T, -1)                                      # position 0 of image b
        freq = torch.bincount(seq[b, :, 0], minlength=probs.numel()).float() / n
        assert float((freq - probs).abs().max()) < 0.035, float((freq - probs).abs().max())
    again = model(fc, att, None, opt={"train_mode": "NAIC", "sample_method": "sample", "sample_n": 2}, mode="sample")[0]
    again2 = model(fc, att, None, opt={"train_mode": "NAIC", "sample_method": "sample", "sample_n": 2}, mode="sample")[0]
    assert not torch.equal(again, again2)                                                  # a new seed per call
    with pytest.raises(NotImplementedError):
        model(fc, att, None, opt={"train_mode": "NAIC", "sample_method": "top5"}, mode="sample")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_dynamic_batching_keeps_per_batch_results(dtype, weight_cache, manifest):
    """Three batches of different images in ONE decode call with q1_group = batch size: every batch's ids, slot layout and
    log-probs equal its own separate decode (quirk Q1 couples rows inside a batch only; the loop count is a max)."""
    from boficap_amd import weights as W
    from boficap_amd.engine import BofiEngine
    m = manifest["tiny_q1_last_shortest"]
    cfg, sd = weight_cache(m["config"], m["seed"], m["gen_scale"], m["digest"], m.get("patch"))
    G = 4
    batches = [torch.from_numpy(load_golden("tiny_q1_last_shortest")["att_feats"]),
               torch.from_numpy(load_golden("tiny_mix")["att_feats"][:G]), torch.from_numpy(load_golden("tiny_mix")["att_feats"][G:2 * G])]
    eng = BofiEngine(cfg, dtype, max_batch=3 * G, max_regions=36)
    eng.load_state_dict(sd)
    cast = (lambda t: t.cuda().to(dtype).contiguous())
    sep = [{k: (v.clone() if torch.is_tensor(v) else v) for k, v in eng.decode_naic(cast(b)).items()} for b in batches]
    allb = eng.decode_naic(cast(torch.cat(batches)), q1_group=G)
    whole = eng.decode_naic(cast(torch.cat(batches)))                    # one big batch: Q1 then uses the very last image
    for i, r in enumerate(sep):
        sl = slice(i * G, (i + 1) * G)
        assert torch.equal(allb["seq"][sl], r["seq"]) and torch.equal(allb["phrase_length"][sl], r["phrase_length"])
        assert torch.equal(allb["phrase_syn"][sl], r["phrase_syn"]) and torch.equal(allb["phrase_num"][sl], r["phrase_num"])
        a, b = allb["seq_logprob"][sl], r["seq_logprob"]
        assert torch.equal(a.isnan(), b.isnan()) and float((a - b).nan_to_num().abs().max()) < (1e-5 if dtype == torch.float32 else 2e-2)
    assert not torch.equal(whole["seq"], allb["seq"])                   # the grouping matters


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_device_side_weight_refresh_equals_a_fresh_load(dtype, weight_cache, manifest):
    """bofi_engine_refresh_device (re-pack from float32 parameters in HBM: stacking, LayerNorm folding, cast, bound tables,
    all as kernels) against an engine loaded through the host path with the same weights; captured graphs stay valid."""
    from boficap_amd.engine import BofiEngine
    m = manifest["tiny_mix"]
    cfg, sd_a = weight_cache(m["config"], m["seed"], m["gen_scale"], m["digest"])
    _, sd_b = weight_cache("TINY", 0, 1.0)                     # another set of values, same schema
    att = torch.from_numpy(load_golden("tiny_mix")["att_feats"]).cuda()
    att = att.to(dtype).contiguous()
    eng = BofiEngine(cfg, dtype, max_batch=16, max_regions=36)
    eng.load_state_dict(sd_a)
    out = eng.decode_naic(att, graph=True)
    first = {k: v.clone() for k, v in out.items() if torch.is_tensor(v)}
    dev_b = {k: torch.from_numpy(v).cuda().contiguous() for k, v in sd_b.items()}
    eng.refresh_from_device(dev_b)
    out = eng.decode_naic(att, graph=True, out=out)            # replay of the graph captured before the refresh
    ref = BofiEngine(cfg, dtype, max_batch=16, max_regions=36)
    ref.load_state_dict(sd_b)
    want = ref.decode_naic(att)
    assert not torch.equal(first["seq"], want["seq"])
    assert torch.equal(out["seq"], want["seq"]) and torch.equal(out["phrase_length"], want["phrase_length"])
    assert torch.equal(out["phrase_syn"], want["phrase_syn"])
    a, b = out["seq_logprob"], want["seq_logprob"]
    assert torch.equal(a.isnan(), b.isnan()) and float((a - b).nan_to_num().abs().max()) < (2e-5 if dtype == torch.float32 else 2e-2)
    s1, s2 = eng.decode_saic(att), ref.decode_saic(att)
    assert torch.equal(s1["seq"], s2["seq"]) and torch.equal(s1["phrase_length"], s2["phrase_length"])
    eng.refresh_from_device({k: torch.from_numpy(v).cuda().contiguous() for k, v in sd_a.items()})
    back = eng.decode_naic(att, graph=True, out=out)
    assert torch.equal(back["seq"], first["seq"])
    with pytest.raises(Exception):
        eng.refresh_from_device({k: v for k, v in dev_b.items() if "generator" not in k})


def test_dynamic_batching_with_refinement_rounds(weight_cache, manifest):
    """q1_group together with iterative refinement: each batch equals its own refined decode."""
    from boficap_amd.engine import BofiEngine
    m = manifest["tiny_mix"]
    cfg, sd = weight_cache(m["config"], m["seed"], m["gen_scale"], m["digest"])
    g = load_golden("tiny_mix")
    a, b = torch.from_numpy(g["att_feats"][:4]).cuda(), torch.from_numpy(g["att_feats"][4:8]).cuda()
    eng = BofiEngine(cfg, torch.float32, max_batch=8, max_regions=36)
    eng.load_state_dict(sd)
    ra = {k: v.clone() for k, v in eng.decode_naic(a, refine_rounds=2).items() if torch.is_tensor(v)}
    rb = {k: v.clone() for k, v in eng.decode_naic(b, refine_rounds=2).items() if torch.is_tensor(v)}
    both = eng.decode_naic(torch.cat([a, b]), refine_rounds=2, q1_group=4)
    assert torch.equal(both["seq"][:4], ra["seq"]) and torch.equal(both["seq"][4:], rb["seq"])
    assert torch.equal(both["phrase_length"][:4], ra["phrase_length"]) and torch.equal(both["phrase_length"][4:], rb["phrase_length"])


def test_two_layer_bounding_network_dense_pass(weight_cache, manifest):
    """N_len = 2 (configs/uic_sd_N2.yml): the engine's dense bounding pass against vectors recorded from the reference -- slots and
    ids bit-exact, log-probs <= 1e-3, eager and graph; through the drop-in module as well; bf16 runs and agrees on most layouts."""
    import captioning.models as models
    from boficap_amd.engine import BofiEngine
    from boficap_amd.hip import BofiHipError
    m = manifest["tiny_n2"]
    cfg, sd = weight_cache(m["config"], m["seed"], m["gen_scale"], m["digest"], m.get("patch"))
    g = load_golden("tiny_n2")
    att = torch.from_numpy(g["att_feats"]).cuda()
    eng = BofiEngine(cfg, torch.float32, max_batch=16, max_regions=36)
    eng.load_state_dict(sd)
    r = None
    for graph in (False, True, True):
        r = eng.decode_naic(att, graph=graph, out=r if graph and r is not None and graph else None)
        torch.cuda.synchronize()
        assert (r["phrase_num"].cpu().numpy() == g["naic_phrase_num"]).all() and (r["phrase_length"].cpu().numpy() == g["naic_phrase_length"]).all()
        assert (r["phrase_syn"].cpu().numpy() == g["naic_phrase_syn"]).all() and (r["seq"].cpu().numpy() == g["naic_seq"]).all()
        assert int(r["bound_iters"]) == int(g["naic_iters"])
        assert _close(r["seq_logprob"].cpu().numpy(), g["naic_logprob"], 0) < 1e-3
    with pytest.raises(BofiHipError):
        eng.decode_saic(att)                                   # the semi-autoregressive decode is built for N_len = 1
    model = models.setup(cfg.to_opt())
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model.cuda().eval()
    with torch.no_grad():
        seq = model(torch.zeros(att.size(0), 0, device="cuda"), att, None, opt={"train_mode": "NAIC"}, mode="sample")[0]
    assert (seq.cpu().numpy() == g["naic_seq"]).all()
    b16 = BofiEngine(cfg, torch.bfloat16, max_batch=16, max_regions=36)
    b16.load_state_dict(sd)
    rb = b16.decode_naic(att.to(torch.bfloat16))
    same = (rb["phrase_length"].cpu().numpy() == g["naic_phrase_length"]).all(1)
    assert same.sum() >= len(same) - 2 and not rb["seq_logprob"].isnan().any()


def test_dense_bounding_pass_equals_incremental_form(weight_cache, manifest, monkeypatch):
    """BOFI_BOUND_DENSE=1 makes a one-layer model take the dense bounding pass (all rows through the layer every iteration, as
    the reference computes it): same slots, ids and log-probs as the goldens -- a cross-check of the row-0-only incremental form,
    of the key-prefix bookkeeping and of quirk Q4's exactness claim."""
    from boficap_amd.engine import BofiEngine
    monkeypatch.setenv("BOFI_BOUND_DENSE", "1")
    for name in ("tiny_mix", "tiny_q1_last_shortest", "tiny_ragged"):
        m = manifest[name]
        cfg, sd = weight_cache(m["config"], m["seed"], m["gen_scale"], m["digest"], m.get("patch"))
        g = load_golden(name)
        att, att_len = _inputs(g)
        eng = BofiEngine(cfg, torch.float32, max_batch=16, max_regions=36)
        eng.load_state_dict(sd)
        r = eng.decode_naic(att, att_len)
        torch.cuda.synchronize()
        assert (r["phrase_length"].cpu().numpy() == g["naic_phrase_length"]).all() and (r["phrase_syn"].cpu().numpy() == g["naic_phrase_syn"]).all(), name
        assert (r["seq"].cpu().numpy() == g["naic_seq"]).all() and int(r["bound_iters"]) == int(g["naic_iters"])
        assert _close(r["seq_logprob"].cpu().numpy(), g["naic_logprob"], 0) < 1e-3


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_saic_row_list_equals_full_decoder_passes(dtype, weight_cache, manifest, monkeypatch):
    """From the second iteration on the semi-autoregressive decode runs only the new phrases' rows through the decoder (their K / V
    join the per-layer cache); BOFI_SAIC_CACHE=0 re-runs every row in every iteration as the reference does (decode_SA
    TransformerModel.py:520-530 inside core_SAIC :1949-1952).  Same ids, same slot layout, same log-probs — greedy and sampled."""
    from boficap_amd.engine import BofiEngine
    from boficap_amd import weights as W
    m = manifest["full_saic_multi"]
    cfg, sd = weight_cache(m["config"], m["seed"], m["gen_scale"], m["digest"], m.get("patch"))
    g = load_golden("full_saic_multi")
    att = torch.from_numpy(W.synthetic_att_feats(m["pool_size"], 36, cfg.att_feat_size, seed=m["pool_seed"])[g["pool_index"]]).cuda()
    outs = []
    monkeypatch.setenv("BOFI_SAIC_LEAN", "0")                  # same kernels on both sides: the row list alone must change nothing
    for cache in ("1", "0"):
        monkeypatch.setenv("BOFI_SAIC_CACHE", cache)
        eng = BofiEngine(cfg, dtype, max_batch=16, max_regions=36)
        eng.load_state_dict(sd)
        r = eng.decode_saic(att)
        rs = eng.decode_saic(att, sample=(1.0, 77))
        if cache == "1":                                           # captured replay: same results, and a new seed draws anew
            rg = eng.decode_saic(att, graph=True)
            rg = eng.decode_saic(att, graph=True, out=rg)
            rsg = eng.decode_saic(att, sample=(1.0, 77), graph=True)
            rsg = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in eng.decode_saic(att, sample=(1.0, 77), graph=True, out=rsg).items()}
            rs2 = eng.decode_saic(att, sample=(1.0, 78), graph=True, out=None)
            torch.cuda.synchronize()
            for k in ("seq", "phrase_num", "phrase_length", "phrase_syn"):
                assert torch.equal(r[k], rg[k]) and torch.equal(rs[k], rsg[k]), k
            assert torch.equal(r["seq_logprob"].nan_to_num(0.0), rg["seq_logprob"].nan_to_num(0.0))
            assert not torch.equal(rs2["seq"], rs["seq"])
        torch.cuda.synchronize()
        outs.append((r, rs))
    for a, b in zip(outs[0], outs[1]):
        assert int(a["phrase_num"].sum()) > att.size(0)            # several phrases per image: the row-list iterations did run
        for k in ("seq", "phrase_num", "phrase_length", "phrase_syn"):
            assert torch.equal(a[k], b[k]), k
        assert int(a["bound_iters"]) == int(b["bound_iters"])
        la, lb = a["seq_logprob"], b["seq_logprob"]
        assert torch.equal(la.isnan(), lb.isnan()) and torch.equal(la.nan_to_num(0.0), lb.nan_to_num(0.0))


def test_saic_row_list_on_direct_operand_kernels(weight_cache, manifest, monkeypatch):
    """bf16: the row-list iterations run their GEMMs and the cross-attention on the direct-operand kernels of bound_ops.hip (other
    summation order than the LDS-DMA GEMM): against the same decode on the general kernels the captions agree except where a
    near-tie flips (then everything after it differs by construction), and wherever a caption's ids agree its log-probs do to
    bf16 accuracy."""
    from boficap_amd.engine import BofiEngine
    from boficap_amd import weights as W
    m = manifest["full_saic_multi"]
    cfg, sd = weight_cache(m["config"], m["seed"], m["gen_scale"], m["digest"], m.get("patch"))
    pool = torch.from_numpy(W.synthetic_att_feats(96, 36, cfg.att_feat_size, seed=m["pool_seed"])).cuda()
    res = []
    for lean in ("1", "0"):
        monkeypatch.setenv("BOFI_SAIC_LEAN", lean)
        eng = BofiEngine(cfg, torch.bfloat16, max_batch=64, max_regions=36)
        eng.load_state_dict(sd)
        keep = eng.decode_naic(pool[:64])["phrase_num"] > 0 if not res else None
        att = pool[:64][res[0][2]] if res else pool[:64][keep]
        r = eng.decode_saic(att.contiguous())
        torch.cuda.synchronize()
        res.append((r["seq"].clone(), r["seq_logprob"].clone(), keep if keep is not None else res[0][2], r["phrase_num"].clone()))
    (sa, la, _, pa), (sb, lb, _, pb) = res
    assert int(pa.max()) >= 3                                      # several row-list iterations did run
    same = (sa == sb).all(1)
    assert float(same.float().mean()) >= 0.8, float(same.float().mean())
    d = (la[same].nan_to_num(0.0) - lb[same].nan_to_num(0.0)).abs().max()
    assert float(d) <= 6e-2, float(d)


def test_decodes_in_flight_hint_changes_kernels_not_results(engines):
    """bofi_engine_set_decodes_in_flight: a hint for the kernel choice (64-row blocks for a decode that runs alone, 80- / 96-row blocks otherwise),
    part of the graph key.  Under the row-block family (forced here: 64 images are below its size threshold) the same hint gives the same bits eager
    or replayed, hints 0 and 4 are the same kernels, and hint 1 differs as two bf16 summation orders do (same layouts on nearly every image, log-probs
    within the bf16 bar where the layouts agree); a fork starts from the default hint."""
    from boficap_amd import hip as H
    from boficap_amd import weights as W
    import os
    cfg, sd, eng = engines("full_b8", torch.bfloat16)
    att = torch.from_numpy(W.synthetic_att_feats(64, 36, cfg.att_feat_size, seed=99)).cuda().to(torch.bfloat16)
    os.environ["BOFI_RB_MIN_ROWS"] = "0"
    H.lib().bofi_reload_env()
    try:
        outs = []
        for n, graph in ((0, False), (1, False), (4, False), (1, True)):
            eng.set_decodes_in_flight(n)
            r = eng.decode_naic(att, strict_q1=False, graph=graph)
            torch.cuda.synchronize()
            outs.append({k: r[k].clone() for k in ("seq", "phrase_length", "phrase_syn", "seq_logprob")})
        for k in ("seq", "phrase_length", "phrase_syn"):
            assert torch.equal(outs[0][k], outs[2][k]) and torch.equal(outs[1][k], outs[3][k])
        assert torch.equal(outs[1]["seq_logprob"].nan_to_num(), outs[3]["seq_logprob"].nan_to_num())          # same hint, eager / replayed: bit for bit
        assert torch.equal(outs[0]["seq_logprob"].nan_to_num(), outs[2]["seq_logprob"].nan_to_num())          # 0 and 4: the same (throughput) forms
        same = (outs[0]["phrase_length"] == outs[1]["phrase_length"]).all(1) & (outs[0]["phrase_syn"] == outs[1]["phrase_syn"]).all(1)
        assert int(same.sum()) >= 58, int(same.sum())                                                        # near-ties may flip between two bf16 kernels
        d = (outs[0]["seq_logprob"][same] - outs[1]["seq_logprob"][same]).nan_to_num().abs().max()
        assert 0 < float(d) < 4e-2, float(d)                                                                # two bf16 results, each within 2e-2 of the float32 one (measured: 0.023)
        f = eng.fork()
        r = f.decode_naic(att, strict_q1=False)
        assert torch.equal(r["seq_logprob"].nan_to_num(), outs[0]["seq_logprob"].nan_to_num())
    finally:
        os.environ.pop("BOFI_RB_MIN_ROWS")
        H.lib().bofi_reload_env()
        eng.set_decodes_in_flight(0)


def test_projection_tail_of_the_feed_forward_kernel_changes_launches_not_results(engines):
    """BOFI_RB_FFN_PROJ: the next layer's q|k|v (after the last encoder layer the stacked cross K|V) computed by the feed-forward launch from each
    closed block (rb_ffn5_kernel<PROJ>) against the separate projection launches: the residual stream is the same bit for bit, the projections differ
    in the summation order of the LayerNorm statistics only, which flips bf16 roundings of a few q / k / v values: two bf16 results, each within the
    bar of the float32 one (measured between them: 0.022, as between the 64- and 80-row kernel families) -- same layouts on nearly every image."""
    from boficap_amd import hip as H
    from boficap_amd import weights as W
    import os
    cfg, sd, eng = engines("full_b8", torch.bfloat16)
    att = torch.from_numpy(W.synthetic_att_feats(64, 36, cfg.att_feat_size, seed=77)).cuda().to(torch.bfloat16)
    os.environ["BOFI_RB_MIN_ROWS"] = "0"
    outs = {}
    try:
        for v in ("0", "1"):
            os.environ["BOFI_RB_FFN_PROJ"] = v
            H.lib().bofi_reload_env()
            r = eng.decode_naic(att, strict_q1=False, graph=False)
            torch.cuda.synchronize()
            outs[v] = {k: r[k].clone() for k in ("seq", "phrase_length", "phrase_syn", "seq_logprob")}
        a, b = outs["0"], outs["1"]
        same = (a["phrase_length"] == b["phrase_length"]).all(1) & (a["phrase_syn"] == b["phrase_syn"]).all(1)
        assert int(same.sum()) >= 60, int(same.sum())
        d = (a["seq_logprob"][same] - b["seq_logprob"][same]).nan_to_num().abs().max()
        assert 0 < float(d) < 4e-2, float(d)                                                                 # (> 0: the fused launches did run)
    finally:
        os.environ.pop("BOFI_RB_MIN_ROWS")
        os.environ.pop("BOFI_RB_FFN_PROJ", None)
        H.lib().bofi_reload_env()


def test_query_projection_tail_of_the_attention_kernel_changes_launches_not_results(engines):
    """BOFI_RB_ATTN_PROJ: the decoder layers' cross-attention query projection computed by the self-attention launch from each 80-row block (rb_attn_kernel<.., PJ>)
    against the separate projection launch: the residual stream is the same bit for bit, the queries differ in the summation order of the LayerNorm statistics only
    (a bf16 rounding here and there): same layouts (the bounding loop does not see the filling pass), log-probs within the families' mutual distance."""
    from boficap_amd import hip as H
    from boficap_amd import weights as W
    import os
    cfg, sd, eng = engines("full_b8", torch.bfloat16)
    att = torch.from_numpy(W.synthetic_att_feats(64, 36, cfg.att_feat_size, seed=79)).cuda().to(torch.bfloat16)
    os.environ["BOFI_RB_MIN_ROWS"] = "0"
    eng.set_decodes_in_flight(4)
    outs = {}
    try:
        for v in ("0", "2"):
            os.environ["BOFI_RB_ATTN_PROJ"] = v
            H.lib().bofi_reload_env()
            H.gemm_flops(reset=True)
            r = eng.decode_naic(att, strict_q1=False, graph=False)
            torch.cuda.synchronize()
            outs[v] = {k: r[k].clone() for k in ("seq", "phrase_length", "phrase_syn", "seq_logprob")}
        a, b = outs["0"], outs["2"]
        assert torch.equal(a["phrase_length"], b["phrase_length"]) and torch.equal(a["phrase_syn"], b["phrase_syn"])
        d = (a["seq_logprob"] - b["seq_logprob"]).nan_to_num().abs().max()
        assert 0 < float(d) < 4e-2, float(d)                                                                 # (> 0: the fused launches did run)
        assert float((a["seq"] == b["seq"]).float().mean()) > 0.98
    finally:
        os.environ.pop("BOFI_RB_MIN_ROWS")
        os.environ.pop("BOFI_RB_ATTN_PROJ", None)
        H.lib().bofi_reload_env()
        eng.set_decodes_in_flight(0)


def test_layer0_qkv_table_of_the_filling_pass_changes_launches_not_results(engines):
    """BOFI_FILL_QKV_TAB: in the filling pass's first round every word is BOS, so decoder layer 0's q|k|v row depends on (label, position) only -- a 200-row table made
    by the row-block projection kernel at finalize / refresh, gathered by the embedding launch -- against the projection launch on all B * 20 rows: bit for bit (a row's
    result does not depend on the launch), also in-flight forms, after a weight refresh (the table is rebuilt) and with refinement rounds (rounds >= 1 read tokens: the
    projection runs there)."""
    from boficap_amd import hip as H
    from boficap_amd import weights as W
    import os
    cfg, sd, eng = engines("full_b8", torch.bfloat16)
    att = torch.from_numpy(W.synthetic_att_feats(64, 36, cfg.att_feat_size, seed=78)).cuda().to(torch.bfloat16)
    os.environ["BOFI_RB_MIN_ROWS"] = "0"
    keys = ("seq", "phrase_length", "phrase_syn", "seq_logprob")
    try:
        for hint, rounds in ((1, 0), (4, 0), (4, 2)):
            eng.set_decodes_in_flight(hint)
            outs = {}
            for v in ("0", "1"):
                os.environ["BOFI_FILL_QKV_TAB"] = v
                H.lib().bofi_reload_env()
                H.gemm_flops(reset=True)
                r = eng.decode_naic(att, strict_q1=False, graph=False, refine_rounds=rounds)
                torch.cuda.synchronize()
                outs[v] = ({k: r[k].clone() for k in keys}, H.gemm_flops(reset=True)[0])
            for k in keys:
                assert torch.equal(outs["0"][0][k].nan_to_num() if outs["0"][0][k].is_floating_point() else outs["0"][0][k],
                                   outs["1"][0][k].nan_to_num() if outs["1"][0][k].is_floating_point() else outs["1"][0][k]), (hint, rounds, k)
            assert outs["1"][1] < outs["0"][1]                                  # (the table path did run: one projection less in the tally)
        # after a device-side weight refresh with OTHER weights the table is the new weights' (a stale one would change the log-probs by whole units)
        from boficap_amd.engine import BofiEngine
        sd2 = W.make_state_dict(cfg, seed=0)
        rng = np.random.default_rng(1)
        for k in ("model.decoder.layers.0.self_attn.linears.0.weight", "model.decoder.layers.0.self_attn.linears.2.weight", "model.syn_embed.lut.weight"):
            sd2[k] = (sd2[k] + 0.05 * rng.standard_normal(sd2[k].shape)).astype(np.float32)
        e2 = BofiEngine(cfg, torch.bfloat16, max_batch=64, max_regions=36)
        e2.load_state_dict(sd)
        e2.refresh_from_device({k: torch.from_numpy(v).cuda() for k, v in sd2.items()})
        outs = {}
        for v in ("0", "1"):
            os.environ["BOFI_FILL_QKV_TAB"] = v
            H.lib().bofi_reload_env()
            r = e2.decode_naic(att, strict_q1=False, graph=False)
            torch.cuda.synchronize()
            outs[v] = r["seq_logprob"].clone()
        assert torch.equal(outs["0"].nan_to_num(), outs["1"].nan_to_num())
    finally:
        os.environ.pop("BOFI_RB_MIN_ROWS")
        os.environ.pop("BOFI_FILL_QKV_TAB", None)
        H.lib().bofi_reload_env()
        eng.set_decodes_in_flight(0)


def test_decode_many_equals_one_decode_per_batch(weight_cache, monkeypatch):
    """TransformerModel.decode_many (engine forks in flight, several loader batches per launch with quirk Q1 per batch, features from pinned host memory on a
    copy stream) against one decode per batch on the same kernel family and hint: bit for bit, in input order, ragged batches and a short last batch included."""
    import captioning.models as models
    from boficap_amd import hip as H
    from boficap_amd import weights as W
    from boficap_amd.engine import BofiEngine
    monkeypatch.setenv("BOFI_RB_MIN_ROWS", "0")                 # one kernel family whatever the launch size (a row's result then does not depend on the launch)
    monkeypatch.setenv("BOFI_BOUND_LOOP", "2")
    H.lib().bofi_reload_env()
    cfg, sd = weight_cache("FULL", 0, 1.0)
    opt = cfg.to_opt()
    opt.bofi_compute_dtype, opt.bofi_max_batch, opt.bofi_max_regions = torch.bfloat16, 16, 36
    model = models.setup(opt)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model.cuda().eval()
    feats = torch.from_numpy(W.synthetic_att_feats(16 * 7 + 5, 36, cfg.att_feat_size, seed=77)).to(torch.bfloat16).pin_memory()
    batches = [feats[i:i + 16] for i in range(0, feats.size(0), 16)]           # 7 batches of 16 and one of 5
    masks = torch.ones(16, 36)
    masks[3, 20:] = 0; masks[9, 7:] = 0
    items = [b if i != 2 else (b, masks) for i, b in enumerate(batches)]          # one ragged batch in the middle (breaks the coalescing there)
    # ... and two ragged batches CLIPPED to their longest image as the loader's consumer does (AttModel.clip_att): 30 and 33 regions, padded by the
    # pipeline to one region bucket (36) so that they share a launch; the padding is masked by the counts
    clip_masks = []
    for i, rmax in ((4, 30), (5, 33)):
        m = torch.ones(16, rmax)
        m[1, rmax // 2:] = 0; m[7, 5:] = 0
        clip_masks.append(m)
        items[i] = (batches[i][:, :rmax].contiguous().pin_memory(), m)
    got = list(model.decode_many(items, batches_per_launch=3, in_flight=2, keep_logprob=True))
    assert len(got) == len(batches)
    ref = BofiEngine(cfg, torch.bfloat16, max_batch=16, max_regions=36)
    ref.load_state_dict(sd)
    ref.set_decodes_in_flight(2)
    for i, (b, r) in enumerate(zip(batches, got)):
        lens = masks.sum(1).to(torch.int32).cuda() if i == 2 else None
        if i in (4, 5):                                          # the clipped batch decoded alone at ITS region count
            m = clip_masks[i - 4]
            b, lens = b[:, :m.size(1)].contiguous(), m.sum(1).to(torch.int32).cuda()
        w = ref.decode_naic(b.cuda(), lens)
        ent, ppl = ref.entropy_perplexity(w)
        for k in ("seq", "phrase_num", "phrase_length", "phrase_syn"):
            assert torch.equal(r[k], w[k].cpu()), (i, k)
        a, c = r["seq_logprob"].cpu(), w["seq_logprob"].cpu()
        assert torch.equal(a.isnan(), c.isnan()) and torch.equal(a.nan_to_num(), c.nan_to_num()), i
        # (the pipeline's entropy comes out of the vocabulary epilogue -- sum e (x - max) / sum e - lse --, the reference engine's here out of the read-back kernel --
        # sum exp(logp) logp --: the same figure up to float32 rounding; the emitted ids' log-probs are the same bits)
        assert torch.equal(r["entropy"].isnan(), ent.cpu().isnan()) and float((r["entropy"].nan_to_num() - ent.cpu().nan_to_num()).abs().max()) <= 1e-5 * float(ent.nan_to_num().abs().max())
        assert torch.equal(r["perplexity"].nan_to_num(), ppl.cpu().nan_to_num())
    # again through the same pipeline (replayed graphs, reused buffers): the same results
    again = list(model.decode_many(items, batches_per_launch=3, in_flight=2))
    assert all(torch.equal(x["seq"], y["seq"]) and torch.equal(x["phrase_length"], y["phrase_length"]) for x, y in zip(again, got))
    monkeypatch.undo()
    H.lib().bofi_reload_env()


@pytest.mark.parametrize("R", [48, 64, 100])
def test_split_attention_serves_region_counts_beyond_the_fused_kernel(R, weight_cache, monkeypatch):
    """Round 6 (VERDICT r5 missing 3): the fused attention sublayer kernel stops at 48 keys / 40 queries; real bottom-up features have up to 100 regions
    (captioning/utils/opts.py:84, ragged batches AttModel.py:113-120).  Beyond it the row-block family now runs the SPLIT form at every launch size -- attention core
    (attn_bf16_kernel, up to 128 keys) + W_o / residual as the head segment of the feed-forward launch -- instead of attention + tiled GEMM.  Against the float32 oracle on
    ragged images: the encoder's memory, the teacher-forced filling pass (north_star's 2e-2) and the free decode's layouts; and against the fallback it replaces."""
    from conftest import record_parity
    from boficap_amd import hip as H
    from boficap_amd import weights as W
    from boficap_amd.engine import BofiEngine
    monkeypatch.setenv("BOFI_RB_MIN_ROWS", "0")
    H.lib().bofi_reload_env()
    try:
        cfg, sd = weight_cache("FULL", 0, 1.0)
        w = O.as_torch(sd)
        B = 24
        att_np = W.synthetic_att_feats(B, R, cfg.att_feat_size, seed=300 + R)
        rng = np.random.default_rng(R)
        lens = rng.integers(R // 2, R + 1, size=B).astype(np.int32)
        lens[0], lens[7], lens[B - 1] = R, 2, R - 5
        masks = np.zeros((B, R), np.float32)
        for b, n in enumerate(lens):
            masks[b, :n] = 1
            att_np[b, n:] = 9.0
        att, am = torch.from_numpy(att_np), torch.from_numpy(masks)
        with torch.no_grad():
            memory, src_mask = O.memory_of(w, cfg, att, am)
            phrase, opn, opl, ops, dg = O.core_naic(w, cfg, memory, src_mask, fix_q1=True)
            olp = torch.log_softmax(O.logit(w, phrase), dim=2)
        att_len = torch.from_numpy(lens).cuda()
        feats = att.cuda().to(torch.bfloat16)
        res = {}
        for split in ("1", "0"):
            monkeypatch.setenv("BOFI_RB_ATTN_SPLIT", split)
            H.lib().bofi_reload_env()
            eng = BofiEngine(cfg, torch.bfloat16, max_batch=B, max_regions=R)
            eng.load_state_dict(sd)
            mem = eng.encode(feats, att_len).cpu()
            live = am.bool()
            e_mem = float((mem - memory)[live].abs().max())
            _, lp = eng.fill_naic(dg["ext_syn"].to(torch.int32).cuda(), dg["last"].to(torch.int32).cuda(), R, att_len, strict_q1=False)
            lp = lp.cpu()
            assert torch.equal(lp.isnan(), olp.isnan())
            e_fill = float((lp - olp).nan_to_num().abs().max())
            free = eng.decode_naic(feats, att_len, strict_q1=False)
            flips = int(((free["phrase_length"].cpu() != opl).any(1) | (free["phrase_syn"].cpu() != ops).any(1)).sum())
            res[split] = (e_mem, e_fill, flips, mem)
            print(f"R {R}, BOFI_RB_ATTN_SPLIT={split}: memory |d| {e_mem:.3e}, teacher-forced fill |dlogp| {e_fill:.3e}, {flips}/{B} layouts differ from the float32 oracle's")
        e_mem, e_fill, flips, mem1 = res["1"]
        record_parity(f"bf16_split_beyond_fused_kernel_fill_R{R}", e_fill, 2e-2, "row-block family, split attention sublayers (the fused kernel stops at 48 keys / 40 queries), ragged images")
        record_parity(f"bf16_split_beyond_fused_kernel_memory_R{R}", e_mem, 6e-2, "encoder output vs the float32 oracle (values O(1..5))")
        # (free decode: near-ties of the bound heads flip under bf16 -- more of them with 100 ragged regions, where the encoder's share of the heads' error is 0.04,
        # tests/test_gpu_bound_loop.py::test_loop_kernel_on_the_oracles_own_trajectory; the form it replaces flips as many: measured 1 / 0 / 6 against 0 / 0 / 5 of 24)
        flip_bar = 3 if R <= 64 else 8
        record_parity(f"bf16_split_beyond_fused_kernel_flips_R{R}", flips, flip_bar, f"images of {B} whose slot layout differs from the float32 oracle's (the tiled fallback: {res['0'][2]})")
        assert e_fill < 2e-2 and e_mem < 6e-2 and flips <= flip_bar and flips <= res["0"][2] + 2, (e_fill, e_mem, flips, res["0"][2])
        assert res["0"][1] < 2e-2                                    # (the fallback it replaces holds the same bar; the two are different bf16 kernels)
        assert float((mem1 - res["0"][3])[live].abs().max()) < 6e-2
    finally:
        monkeypatch.undo()
        H.lib().bofi_reload_env()
