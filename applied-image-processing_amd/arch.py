"""Architecture tables for the AdaIN encoder/decoder, expressed as data.

Mirrors the module order of the reference's two ``nn.Sequential`` singletons
(reference: Style_3DGS/AdaIN/net.py:6-36 decoder, :38-92 vgg) so that a reference
``state_dict`` (keys ``"<module index>.weight"`` / ``".bias"``) loads unchanged.

Each entry is ``(kind, *params)``:
  ("conv", cin, cout, k)   k = 1 or 3; k=3 convs are preceded by a ("pad",) reflection pad of 1
  ("relu",) ("pad",) ("pool",) ("up",)
``pool`` = MaxPool2d(2, 2, ceil_mode=True); ``up`` = nearest 2x upsample.
"""

_VGG_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512]
_DEC_CFG = [256, "U", 256, 256, 256, 128, "U", 128, 64, "U", 64, 3]

# The encoder is cut after relu4_1: the first 31 modules (reference test.py:185, :38).
ENCODER_CUT = 31


def _build_vgg():
    mods = [("conv", 3, 3, 1)]
    cin = 3
    for v in _VGG_CFG:
        if v == "M":
            mods.append(("pool",))
        else:
            mods += [("pad",), ("conv", cin, v, 3), ("relu",)]
            cin = v
    return mods


def _build_decoder():
    mods = []
    cin = 512
    last = len(_DEC_CFG) - 1
    for i, v in enumerate(_DEC_CFG):
        if v == "U":
            mods.append(("up",))
        else:
            mods += [("pad",), ("conv", cin, v, 3)]
            if i != last:
                mods.append(("relu",))
            cin = v
    return mods


VGG_MODULES = _build_vgg()          # 53 modules
DECODER_MODULES = _build_decoder()  # 29 modules
assert len(VGG_MODULES) == 53 and len(DECODER_MODULES) == 29


def conv_indices(mods):
    """Module indices that carry parameters (the state_dict key prefixes)."""
    return [i for i, m in enumerate(mods) if m[0] == "conv"]


def encoder_plan():
    """Fused layer plan for conv0 .. relu4_1.

    Returns a list of dicts: ``idx`` (state_dict index), ``cin``, ``cout``, ``k``, ``relu``,
    ``src`` in {"direct", "pool"} — "pool" means the conv reads the ceil-mode 2x2 max-pool of
    the previous activation (the pool is fused into the consumer's gather).
    """
    plan, pending = [], "direct"
    for i, m in enumerate(VGG_MODULES[:ENCODER_CUT]):
        if m[0] == "conv":
            nxt = VGG_MODULES[i + 1][0] if i + 1 < ENCODER_CUT else None
            plan.append(dict(idx=i, cin=m[1], cout=m[2], k=m[3], relu=(nxt == "relu"), src=pending))
            pending = "direct"
        elif m[0] == "pool":
            pending = "pool"
    return plan


def decoder_plan():
    """Fused layer plan for the decoder; ``src`` in {"direct", "up"} ("up" = the conv reads the
    nearest-2x upsample of the previous activation, fused into the consumer's gather)."""
    plan, pending = [], "direct"
    n = len(DECODER_MODULES)
    for i, m in enumerate(DECODER_MODULES):
        if m[0] == "conv":
            nxt = DECODER_MODULES[i + 1][0] if i + 1 < n else None
            plan.append(dict(idx=i, cin=m[1], cout=m[2], k=m[3], relu=(nxt == "relu"), src=pending))
            pending = "direct"
        elif m[0] == "up":
            pending = "up"
    return plan


def encoded_size(h, w):
    """Spatial size of relu4_1 for an ``h x w`` image: three ceil-mode halvings."""
    for _ in range(3):
        h, w = (h + 1) // 2, (w + 1) // 2
    return h, w


# Algorithmic work model: 2 flop per MAC, convolutions only.
def conv_flops_encoder(h, w):
    total = 0
    for L in encoder_plan():
        if L["src"] == "pool":
            h, w = (h + 1) // 2, (w + 1) // 2
        total += 2 * h * w * L["cin"] * L["cout"] * L["k"] * L["k"]
    return total


def conv_flops_decoder(hc, wc):
    total = 0
    h, w = hc, wc
    for L in decoder_plan():
        if L["src"] == "up":
            h, w = 2 * h, 2 * w
        total += 2 * h * w * L["cin"] * L["cout"] * L["k"] * L["k"]
    return total


def wino4_geometry(sizes):
    """Tile geometry the F(4,3) x F(2,3) launcher picks for a launch over feature maps ``sizes`` = [(n, h, w), ...] (the segments of
    one launch): 0 = 8 x 32 output pixels per workgroup tile, 1 = 16 x 16 - the one that needs fewer tiles, the default keeping
    near-ties (mirrors ``pick_geo`` in csrc/conv_wino4.hip; tests use it to know which layout a shape exercises)."""
    def tiles(th, tw):
        return sum(n * (-(-h // th)) * (-(-w // tw)) for n, h, w in sizes)

    t0, t1 = tiles(8, 32), tiles(16, 16)
    return 1 if t1 * 101 < t0 * 100 else 0
