"""GPU parity of the CLIP text encoder on the HIP path (flash_diffusion_amd/clip.py) against the REAL third-party
implementation the reference's conditioner wraps: transformers' ``CLIPTextModel`` / ``CLIPTextModelWithProjection``
(embedders/clip/clip_embedder_model.py:10-201), installed in this image -- random-init weights of the architecture (no network),
identical state_dict loaded into both.  Outputs compared: last_hidden_state, pooler_output, hidden_states[-2] (the "hidden" /
clip-skip layer), text_embeds; and the conditioner wrapper's layer selection / zeroing.

Tolerances (stated): fp32 validation mode 1e-4 relative; bf16 production mode 2e-2 (twelve pre-LN layers of bf16 GEMMs)."""
import pytest
import torch

from tests.golden_util import parity_log, rel_err
from tests.isolate import run_isolated

pytestmark = pytest.mark.gpu

CASES = {
    "tiny_quick_gelu_proj": dict(vocab_size=1000, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                                 projection_dim=32, hidden_act="quick_gelu", eos_token_id=2),
    "tiny_gelu_eos": dict(vocab_size=1000, hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=2,
                          projection_dim=None, hidden_act="gelu", eos_token_id=999),
    "clip_l": dict(vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12,
                   projection_dim=None, hidden_act="quick_gelu", eos_token_id=2),          # openai/clip-vit-large-patch14 text tower
}


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
@pytest.mark.parametrize("name", list(CASES))
def test_clip_text_encoder_matches_transformers(name, precision):
    run_isolated(__name__, "_body", (name, precision), timeout=600)


def _hf(kw):
    from transformers import CLIPTextConfig, CLIPTextModel, CLIPTextModelWithProjection
    kw = dict(kw)
    proj = kw.pop("projection_dim")
    cfg = CLIPTextConfig(max_position_embeddings=77, projection_dim=proj or 512, bos_token_id=0, pad_token_id=1, **kw)
    torch.manual_seed(0)
    m = (CLIPTextModelWithProjection if proj else CLIPTextModel)(cfg).eval()
    with torch.no_grad():   # transformers' default init (std 0.02) leaves the residual stream tiny: scale up so every term matters
        for n, p in m.named_parameters():
            if p.dim() > 1 and "embedding" not in n:
                p.mul_(3.0)
            elif p.dim() == 1 and "bias" in n:
                p.add_(0.05 * torch.randn(p.shape))
    return m


def _body(name, precision):
    from flash_diffusion_amd.clip import MiClipEmbedder, MiCLIPTextModel
    kw = CASES[name]
    hf = _hf(kw)
    mi = MiCLIPTextModel(**kw, precision=precision)
    sd = {(k if k.startswith(("text_model.", "text_projection.")) else "text_model." + k): v for k, v in hf.state_dict().items()}
    sd = {k: v for k, v in sd.items() if not k.endswith("position_ids")}
    missing = mi.load_state_dict(sd, strict=True)
    mi = mi.cuda()
    mi.freeze()
    g = torch.Generator().manual_seed(1)
    B, V = 3, kw["vocab_size"]
    ids = torch.randint(3, V - 2, (B, 77), generator=g)
    eos = kw["eos_token_id"] if kw["eos_token_id"] != 2 else V - 1          # legacy id 2: the pooled row is the arg-max id
    for b, pos in enumerate((5, 40, 76)):
        ids[b, pos] = eos
        ids[b, pos + 1:] = 1
    with torch.no_grad():
        ref = hf(input_ids=ids, output_hidden_states=True)
    out = mi(ids.cuda(), output_hidden_states=True)
    errs = {"last": rel_err(out.last_hidden_state, ref.last_hidden_state),
            "hidden[-2]": rel_err(out.hidden_states[-2], ref.hidden_states[-2]), "hidden[0]": rel_err(out.hidden_states[0], ref.hidden_states[0])}
    pooled_ref = ref.pooler_output if hasattr(ref, "pooler_output") and ref.pooler_output is not None else None
    if pooled_ref is None:      # ...WithProjection returns text_embeds; its pooled row is the text model's
        pooled_ref = ref.last_hidden_state[torch.arange(B), ids.argmax(-1) if kw["eos_token_id"] == 2 else (ids == eos).int().argmax(-1)]
    errs["pooled"] = rel_err(out.pooler_output, pooled_ref)
    if kw["projection_dim"]:
        errs["text_embeds"] = rel_err(out.text_embeds, ref.text_embeds)
    parity_log(f"clip text encoder {name} [{precision}]: " + " ".join(f"{k}={v:.3e}" for k, v in errs.items()), "nets_parity.txt")
    assert len(out.hidden_states) == len(ref.hidden_states) == kw["num_hidden_layers"] + 1
    tol = 1e-4 if precision == "fp32" else 2e-2
    assert all(v <= tol for v in errs.values()), errs
    # the conditioner wrapper (clip_embedder_model.py:47-104): layer selection, pooled vector, zeroing
    tok = lambda texts: ids[:len(texts)]
    e = MiClipEmbedder(mi, tok, layer="hidden", layer_idx=-2, always_return_pooled=True)
    o = e({"text": ["a", "b", "c"]})
    assert set(o) == {"crossattn", "vector"} and torch.equal(o["crossattn"], out.hidden_states[-2])
    z = e({"text": ["a", "b", "c"]}, force_zero_embedding=True)
    assert float(z["crossattn"].abs().max()) == 0.0 and float(z["vector"].abs().max()) == 0.0
    assert MiClipEmbedder(mi, tok, layer="pooled")({"text": ["a", "b", "c"]})["crossattn"].shape[1] == 1
