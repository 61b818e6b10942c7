"""Static architecture tables for MVFNet-ResNet50/101/152 (no torch needed).

One source of truth for (a) the state_dict key layout the reference produces and released
checkpoints carry (`backbone.layerL.B.conv1.{net,shift_conv,h_conv,w_conv,bn}`),
(b) the per-layer conv shapes the HIP engine plans its launches from, (c) FLOP accounting.

Reference: codes/models/backbones/resnet.py:357-363 (arch_settings), :247-326 (make_res_layer),
codes/models/modules/MVF.py:18-49 (which blocks get an MVF), :55-89 (its parameters).
"""

STAGE_BLOCKS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}


def _bn_entries(prefix, n):
    return {prefix + "weight": (n,), prefix + "bias": (n,), prefix + "running_mean": (n,),
            prefix + "running_var": (n,), prefix + "num_batches_tracked": ()}


def blocks(depth):
    """Yield dicts describing every bottleneck: name prefix, inplanes, planes, stride, has_down, stage."""
    inplanes = 64
    for li, nblk in enumerate(STAGE_BLOCKS[depth]):
        planes = 64 << li
        for bi in range(nblk):
            stride = 2 if (bi == 0 and li > 0) else 1
            has_down = bi == 0
            yield dict(prefix="layer%d.%d." % (li + 1, bi), stage=li, index=bi, inplanes=inplanes,
                       planes=planes, stride=stride, has_down=has_down)
            inplanes = planes * 4


def state_dict_shapes(depth=50, alpha=0.125, mvf_freq=(0, 0, 1, 1), mode="THW", share=False,
                      num_classes=400, with_head=True, backbone_prefix="backbone.", head_prefix="cls_head."):
    """key -> shape for the full recognizer, identical to the reference model's state_dict()."""
    bp = backbone_prefix
    sd = {bp + "conv1.weight": (64, 3, 7, 7)}
    sd.update(_bn_entries(bp + "bn1.", 64))
    for b in blocks(depth):
        p = bp + b["prefix"]
        cin, pl = b["inplanes"], b["planes"]
        cs = int(cin * alpha) if mvf_freq[b["stage"]] else 0
        if mvf_freq[b["stage"]]:
            sd[p + "conv1.net.weight"] = (pl, cin, 1, 1)
            if cs:
                sd[p + "conv1.shift_conv.weight"] = (cs, 1, 3, 1, 1)
                sd.update(_bn_entries(p + "conv1.bn.", cs))
                if not share and mode in ("TH", "THW"):
                    sd[p + "conv1.h_conv.weight"] = (cs, 1, 1, 3, 1)
                if not share and mode == "THW":
                    sd[p + "conv1.w_conv.weight"] = (cs, 1, 1, 1, 3)
        else:
            sd[p + "conv1.weight"] = (pl, cin, 1, 1)
        sd[p + "conv2.weight"] = (pl, pl, 3, 3)
        sd[p + "conv3.weight"] = (pl * 4, pl, 1, 1)
        sd.update(_bn_entries(p + "bn1.", pl))
        sd.update(_bn_entries(p + "bn2.", pl))
        sd.update(_bn_entries(p + "bn3.", pl * 4))
        if b["has_down"]:
            sd[p + "downsample.0.weight"] = (pl * 4, cin, 1, 1)
            sd.update(_bn_entries(p + "downsample.1.", pl * 4))
    if with_head:
        sd[head_prefix + "new_fc.weight"] = (num_classes, 2048)
        sd[head_prefix + "new_fc.bias"] = (num_classes,)
    return sd


def conv_macs_per_image(depth=50, hw=224, num_classes=400):
    """Multiply-accumulates of conv + fc per IMAGE (x T for a clip). R50 @224: 4.090 G (SURVEY.md 8d:
    32.72 GMAC per 8-frame clip); the MVF depthwise taps (0.05 %) are not counted, as in BASELINE.md."""
    h = (hw + 1) // 2                      # stem 7x7/2 pad 3
    macs = h * h * 64 * 147
    h = (h + 1) // 2                       # maxpool 3x3/2 pad 1
    for b in blocks(depth):
        cin, pl, s = b["inplanes"], b["planes"], b["stride"]
        ho = (h - 1) // s + 1
        macs += h * h * cin * pl           # conv1 1x1 (stride on conv2, style='pytorch')
        macs += ho * ho * pl * pl * 9      # conv2 3x3
        macs += ho * ho * pl * pl * 4      # conv3 1x1
        if b["has_down"]:
            macs += ho * ho * cin * pl * 4
        h = ho
    return macs + 2048 * num_classes


def fused_activation_elems_per_image(depth=50, hw=224):
    """conv-input + conv-output ELEMENTS per image = the activation traffic of a perfectly fused network, every conv input read
    and every conv output written exactly once (85.3 M + 88.9 M elements per 8-frame R50 clip at 224^2 =
    348 MB in bf16)."""
    h = (hw + 1) // 2
    n_in, n_out = hw * hw * 3, h * h * 64
    h = (h + 1) // 2
    for b in blocks(depth):
        cin, pl, s = b["inplanes"], b["planes"], b["stride"]
        ho = (h - 1) // s + 1
        n_in += h * h * cin + h * h * pl + ho * ho * pl
        n_out += h * h * pl + ho * ho * pl + ho * ho * pl * 4
        if b["has_down"]:
            n_in += h * h * cin
            n_out += ho * ho * pl * 4
        h = ho
    return n_in + n_out
