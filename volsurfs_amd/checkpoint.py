"""Reference checkpoints <-> the stacked parameter tensors of NeuralTextureBank.

The reference saves one `state_dict` per model (`<ckpt>/<iter:07d>/models/{rgb_i, alpha_i}.pt`,
/root/reference/volsurfs_py/methods/base_method.py:118-211).  For the neural-texture branch a
model is an `SHNeuralTextures` (models/sh_neural_textures.py:8-62) = ModuleList
`neural_textures[d]` of `NeuralTexture` (models/neural_texture.py:63-79), each holding
`encoding = tcnn.Encoding(HashGrid 16x2, 2^15, base 16, x1.5)` and `network =
tcnn.Network(FullyFusedMLP 32-64-64-C)`; both tcnn modules own ONE flat fp32 `params` tensor
and are also registered a second time inside `model = Sequential(encoding, network)`, so a
reference state_dict carries, per degree d:

    neural_textures.{d}.encoding.params   [708 368]            (= model.0.params)
    neural_textures.{d}.network.params    [2048 + 4096 + 64*pad16(C)]   (= model.1.params)

STATED ASSUMPTION (tiny-cuda-nn is an unpinned pip dependency, absent here; `oracle/tcnn_like.py`
uses the same order): the grid's `params` is level-major, entry-major, feature-minor
(`grid[(offset_l + index) * 2 + f]`), i.e. exactly `tables[x].reshape(-1)`; the network's `params`
is the layers' weight matrices concatenated, each row-major `[out, in]`, the last one with its
rows padded to a multiple of 16 — i.e. the first `6144 + 64*pad16(C)` elements of `weights[x]`
(W1[64,32] | W2[64,64] | W3[32,64]; the rows >= C of W3 never reach an output: tiny-cuda-nn
keeps them as padding, here they are loaded as stored and never read).
"""
import torch

from .neural_textures import MAX_DEG, WEIGHTS_PER_TEX

_ENC = ("neural_textures.{d}.encoding.params", "neural_textures.{d}.model.0.params")
_NET = ("neural_textures.{d}.network.params", "neural_textures.{d}.model.1.params")


def is_reference_state_dict(sd):
    return any(k.startswith("neural_textures.") for k in sd)


def _pad16(c):
    return (c + 15) // 16 * 16


@torch.no_grad()
def load_reference_state_dict(bank, shell, typ, sd, strict=True):
    """Copy the parameters of the reference model `rgb_<shell>` (typ 0) or `alpha_<shell>`
    (typ 1) from its state_dict into bank.tables / bank.weights.  Returns the degrees loaded.
    The caller refreshes the f16 copies (`bank.refresh_half_params()`)."""
    E2 = bank.n_entries * 2
    loaded = []
    for d in range(MAX_DEG):
        x = bank.tex_index(shell, typ, d)
        C = bank.tex_channels(x)
        enc = next((sd[k.format(d=d)] for k in _ENC if k.format(d=d) in sd), None)
        net = next((sd[k.format(d=d)] for k in _NET if k.format(d=d) in sd), None)
        if C == 0:
            if strict and (enc is not None or net is not None):
                raise ValueError(f"state_dict holds SH degree {d} but the bank's model has none")
            continue
        if enc is None or net is None:
            if strict:
                raise KeyError(f"reference state_dict lacks neural_textures.{d}.* "
                               f"(shell {shell}, {'alpha' if typ else 'rgb'})")
            continue
        n_net = 2048 + 4096 + 64 * _pad16(C)
        if enc.numel() != E2:
            raise ValueError(f"neural_textures.{d}.encoding.params has {enc.numel()} elements, "
                             f"expected {E2} (16 levels x 2 features, 2^15, base 16, x1.5)")
        if net.numel() != n_net:
            raise ValueError(f"neural_textures.{d}.network.params has {net.numel()} elements, "
                             f"expected {n_net} (32-64-64-pad16({C}))")
        bank.tables[x].copy_(enc.reshape(bank.n_entries, 2).to(bank.tables))
        w = torch.zeros(WEIGHTS_PER_TEX, dtype=bank.weights.dtype, device=bank.weights.device)
        w[:n_net] = net.reshape(-1).to(w)
        w[6144 + 64 * C:] = 0          # padding rows of the output layer: outputs that do not exist
        bank.weights[x].copy_(w)
        loaded.append(d)
    return loaded


@torch.no_grad()
def to_reference_state_dict(bank, shell, typ):
    """The inverse: a state_dict with the reference's keys for model `rgb_<shell>` / `alpha_<shell>`."""
    sd = {}
    for d in range(MAX_DEG):
        x = bank.tex_index(shell, typ, d)
        C = bank.tex_channels(x)
        if C == 0:
            continue
        enc = bank.tables[x].detach().reshape(-1).float().cpu().clone()
        net = bank.weights[x].detach()[:2048 + 4096 + 64 * _pad16(C)].float().cpu().clone()
        for k in _ENC:
            sd[k.format(d=d)] = enc
        for k in _NET:
            sd[k.format(d=d)] = net
    return sd
