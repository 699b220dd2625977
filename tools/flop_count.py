"""Algorithmic FLOP counters (multiply-add = 2 FLOPs) for the secondary bench workloads — SURVEY.md §8(d) asks for the UNet / VAE figures to be
"counted from the layer table at build time".  Pure arithmetic over a config object's fields (duck-typed: the product models' `.config` and the
oracle's dataclasses both fit), no tensors, no model code: bench.py's non-oracle legs and the oracle modules share this one definition.
"""


def unet_flops_fwd(cfg, H: int, W: int, ctx_len: int = 77) -> float:
    """multiply-add = 2 FLOPs; conv 2*k*k*Cin*Cout*H*W, linears 2*M*N*K, attention 4*Sq*Sk*C; per image (SURVEY.md §8d asks for this counter)"""
    fl = 0.0
    ch = cfg.block_out_channels
    nb = len(ch)

    def conv(cin, cout, h, w, k=3):
        return 2.0 * k * k * cin * cout * h * w

    def res(cin, cout, h, w):
        return conv(cin, cout, h, w) + conv(cout, cout, h, w) + (conv(cin, cout, h, w, 1) if cin != cout else 0.0) + 2.0 * 4 * ch[0] * cout

    def tr(c, h, w, n):
        s = h * w
        per = 2.0 * s * c * c * 4 + 4.0 * s * s * c + 2.0 * s * c * c * 2 + 2.0 * ctx_len * cfg.cross_attention_dim * c * 2 + 4.0 * s * ctx_len * c \
            + 2.0 * s * c * 8 * c + 2.0 * s * 4 * c * c
        return n * per + 2 * 2.0 * s * c * c

    h, w = H, W
    fl += conv(cfg.in_channels, ch[0], h, w)
    skip_ch = [ch[0]]
    cin = ch[0]
    for i, typ in enumerate(cfg.down_block_types):
        for j in range(cfg.layers_per_block):
            fl += res(cin, ch[i], h, w)
            cin = ch[i]
            if typ.startswith("CrossAttn"):
                fl += tr(cin, h, w, cfg.transformer_layers_per_block[i])
            skip_ch.append(cin)
        if i < nb - 1:
            h, w = h // 2, w // 2
            fl += conv(cin, cin, h, w)
            skip_ch.append(cin)
    fl += 2 * res(cin, cin, h, w) + tr(cin, h, w, cfg.transformer_layers_per_block[-1])
    for i, typ in enumerate(cfg.up_block_types):
        ri = nb - 1 - i
        for j in range(cfg.layers_per_block + 1):
            fl += res(cin + skip_ch.pop(), ch[ri], h, w)
            cin = ch[ri]
            if typ.startswith("CrossAttn"):
                fl += tr(cin, h, w, cfg.transformer_layers_per_block[ri])
        if i < nb - 1:
            h, w = 2 * h, 2 * w
            fl += conv(cin, cin, h, w)
    fl += conv(cin, cfg.out_channels, h, w)
    return fl


def vae_encoder_flops(cfg, H: int, W: int) -> float:
    """forward FLOPs per image (multiply-add = 2)"""
    def conv(ci, co, h, w, k=3):
        return 2.0 * k * k * ci * co * h * w
    ch = cfg.block_out_channels
    fl = conv(cfg.in_channels, ch[0], H, W)
    cin, h, w = ch[0], H, W
    for i, co in enumerate(ch):
        for j in range(cfg.layers_per_block):
            fl += conv(cin, co, h, w) + conv(co, co, h, w) + (conv(cin, co, h, w, 1) if cin != co else 0)
            cin = co
        if i < len(ch) - 1:
            h, w = h // 2, w // 2
            fl += conv(cin, cin, h, w)
    s = h * w
    fl += 4 * conv(cin, cin, h, w) + 2.0 * s * cin * cin * 4 + 4.0 * s * s * cin
    fl += conv(cin, 2 * cfg.latent_channels, h, w)
    return fl


def pixart_flops_fwd(cfg, H: int, W: int, ctx_len: int = 300, n_ctrl: int = 0) -> float:
    D = cfg.num_attention_heads * cfg.attention_head_dim
    S = (H // cfg.patch_size) * (W // cfg.patch_size)
    blk = 2.0 * S * (4 * D * D + 2 * D * D + 8 * D * D) + 2.0 * ctx_len * 2 * D * D + 4.0 * S * S * D + 4.0 * S * ctx_len * D
    return (cfg.num_layers + n_ctrl) * blk + n_ctrl * 2.0 * S * D * D + 2.0 * S * D * (cfg.patch_size ** 2) * (cfg.in_channels + cfg.out_channels)
