"""GPU parity of the T5 text encoder on the HIP path (flash_diffusion_amd/t5.py) against the REAL third-party
implementation the reference's conditioner wraps: transformers' ``T5EncoderModel`` (embedders/t5/t5_embedder_model.py:11-104),
installed in this image -- random-init weights of the architecture (no network), identical state_dict loaded into both.  Outputs
compared: last_hidden_state, hidden_states[0] (embeddings), hidden_states[-2] (input of the last block), with a padded attention mask;
and the conditioner wrapper's layer selection / mask output / zeroing.  Also the op-level kernels the encoder adds (T5LayerNorm,
biased attention, element-wise product).

Tolerances (stated): fp32 validation mode 1e-4 relative; bf16 production mode 2e-2 (pre-norm blocks of bf16 GEMMs)."""
import pytest
import torch

from tests.golden_util import parity_log, rel_err
from tests.isolate import run_isolated

pytestmark = pytest.mark.gpu

CASES = {
    "tiny_gated": dict(vocab_size=1000, d_model=64, d_kv=16, d_ff=128, num_layers=2, num_heads=4, feed_forward_proj="gated-gelu"),
    "tiny_relu": dict(vocab_size=500, d_model=128, d_kv=32, d_ff=256, num_layers=3, num_heads=2, feed_forward_proj="relu",
                      relative_attention_num_buckets=16, relative_attention_max_distance=20),
    # T5 v1.1-XXL's block shape at 1/4 of its width and 4 of its 24 blocks (head dim 64, 120 tokens, gated feed-forward)
    "xxl_quarter": dict(vocab_size=32128, d_model=1024, d_kv=64, d_ff=2560, num_layers=4, num_heads=16, feed_forward_proj="gated-gelu"),
}


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
@pytest.mark.parametrize("name", list(CASES))
def test_t5_encoder_matches_transformers(name, precision):
    run_isolated(__name__, "_body", (name, precision), timeout=600)


def _hf(kw):
    from transformers import T5Config, T5EncoderModel
    cfg = T5Config(dropout_rate=0.0, **kw)
    torch.manual_seed(0)
    m = T5EncoderModel(cfg).eval()
    with torch.no_grad():   # the default init leaves some terms tiny / the norm weights at exactly 1: make every term matter
        for n, p in m.named_parameters():
            if "layer_norm" in n:
                p.add_(0.2 * torch.randn(p.shape))
            elif "relative_attention_bias" in n:
                p.mul_(4.0)
    return m


def _body(name, precision):
    from flash_diffusion_amd.t5 import MiT5EncoderModel, MiT5TextEmbedder
    kw = CASES[name]
    hf = _hf(kw)
    mi = MiT5EncoderModel(**kw, precision=precision)
    mi.load_state_dict(hf.state_dict(), strict=True)
    mi = mi.cuda()
    mi.freeze()
    g = torch.Generator().manual_seed(1)
    B, S, V = 3, 120, kw["vocab_size"]
    ids = torch.randint(2, V, (B, S), generator=g)
    mask = torch.ones(B, S, dtype=torch.long)
    for b, n in enumerate((7, 64, 120)):      # padded prompts (pad id 0, eos id 1 as T5Tokenizer lays them out)
        ids[b, n - 1] = 1
        ids[b, n:] = 0
        mask[b, n:] = 0
    with torch.no_grad():
        ref = hf(input_ids=ids, attention_mask=mask, output_hidden_states=True)
    out = mi(ids.cuda(), attention_mask=mask.cuda(), output_hidden_states=True)
    assert len(out.hidden_states) == len(ref.hidden_states) == kw["num_layers"] + 1
    valid = mask.bool()[..., None]            # padded query rows carry no information downstream (the denoisers mask them as keys)
    errs = {"last": rel_err(out.last_hidden_state.cpu() * valid, ref.last_hidden_state * valid),
            "last(all rows)": rel_err(out.last_hidden_state, ref.last_hidden_state),
            "hidden[-2]": rel_err(out.hidden_states[-2].cpu() * valid, ref.hidden_states[-2] * valid),
            "hidden[0]": rel_err(out.hidden_states[0], ref.hidden_states[0])}
    parity_log(f"t5 encoder {name} [{precision}]: " + " ".join(f"{k}={v:.3e}" for k, v in errs.items()), "nets_parity.txt")
    tol = 1e-4 if precision == "fp32" else 2e-2
    assert all(v <= tol for v in errs.values()), errs
    # without a mask (transformers: every key visible)
    with torch.no_grad():
        ref2 = hf(input_ids=ids).last_hidden_state
    e2 = rel_err(mi(ids.cuda()).last_hidden_state, ref2)
    assert e2 <= tol, e2
    # the conditioner (t5_embedder_model.py:51-104): layer selection, attention mask output, zeroing
    tok = lambda texts: (ids[:len(texts)], mask[:len(texts)])
    e = MiT5TextEmbedder(mi, tok, layer="hidden", layer_idx=-2, returns_attention_mask=True)
    o = e({"text": ["a", "b", "c"]})
    assert set(o) == {"crossattn", "attention_mask"} and torch.equal(o["crossattn"], out.hidden_states[-2])
    assert torch.equal(o["attention_mask"].cpu(), mask)
    z = e({"text": ["a", "b", "c"]}, force_zero_embedding=True)
    assert float(z["crossattn"].abs().max()) == 0.0 and int(z["attention_mask"].abs().max()) == 0
    assert set(MiT5TextEmbedder(mi, tok)({"text": ["a", "b"]})) == {"crossattn"}


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_t5_kernels(dtype):
    """T5LayerNorm, the biased attention and the element-wise product against torch (fp32 math on the stored values)"""
    from flash_diffusion_amd import ops
    g = torch.Generator().manual_seed(0)
    tol = 1e-5 if dtype == torch.float32 else 2 ** -7
    x = (torch.randn(37, 200, generator=g) * 3).to(dtype)
    w = torch.randn(200, generator=g)
    ref = x.float() * torch.rsqrt(x.float().pow(2).mean(-1, keepdim=True) + 1e-6) * w
    assert rel_err(ops.rmsnorm(x.cuda(), w.cuda(), 1e-6), ref) <= tol
    a, b = torch.randn(1000, generator=g).to(dtype), torch.randn(1000, generator=g).to(dtype)
    assert rel_err(ops.mul(a.cuda(), b.cuda()), a.float() * b.float()) <= tol
    B, H, Sq, Skv, d = 2, 3, 20, 33, 16
    q, k, v = (torch.randn(B, n, H * d, generator=g).to(dtype) for n in (Sq, Skv, Skv))
    bias = torch.randn(H, Sq, Skv, generator=g)
    kb = torch.zeros(B, Skv)
    kb[1, 25:] = torch.finfo(torch.float32).min
    qh, kh, vh = (t.float().view(B, -1, H, d).transpose(1, 2) for t in (q, k, v))
    p = (0.5 * qh @ kh.transpose(-1, -2) + bias[None] + kb[:, None, None, :]).softmax(-1)
    ref = (p @ vh).transpose(1, 2).reshape(B, Sq, H * d)
    out = ops.attn_bias_fwd(q.cuda(), k.cuda(), v.cuda(), H, 0.5, bias.cuda(), kb.cuda())
    assert rel_err(out, ref) <= (1e-5 if dtype == torch.float32 else 2 ** -7), rel_err(out, ref)
    assert rel_err(ops.attn_bias_fwd(q.cuda(), k.cuda(), v.cuda(), H, 0.5), ((0.5 * qh @ kh.transpose(-1, -2)).softmax(-1) @ vh)
                   .transpose(1, 2).reshape(B, Sq, H * d)) <= (1e-5 if dtype == torch.float32 else 2 ** -7)
