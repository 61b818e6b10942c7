"""Case tables shared by tests/golden/make_golden.py (writer) and tests/ (readers). Data only."""

MVF_CASES = [
    # name,            N, T, C,  H, W, alpha, mode,  share, use_hs, planes
    ("thw_small",      2, 4, 32, 6, 5, 0.125, "THW", False, True, 16),
    ("thw_a50",        1, 3, 8,  4, 7, 0.5,   "THW", False, True, 8),
    ("th_small",       2, 4, 16, 5, 6, 0.25,  "TH",  False, True, 8),
    ("t_small",        2, 5, 16, 3, 4, 0.25,  "T",   False, True, 8),
    ("thw_share",      2, 4, 16, 5, 5, 0.25,  "THW", True,  True, 8),
    ("th_share",       1, 4, 16, 6, 6, 0.25,  "TH",  True,  True, 8),
    ("thw_nohs",       2, 4, 16, 5, 6, 0.25,  "THW", False, False, 8),
    ("thw_t1",         3, 1, 16, 4, 4, 0.25,  "THW", False, True, 8),
    ("thw_h1w1",       2, 4, 16, 1, 1, 0.25,  "THW", False, True, 8),
    ("thw_zero_cs",    2, 4, 4,  4, 4, 0.125, "THW", False, True, 4),   # int(4*0.125)=0 -> pass-through
    ("thw_l3_shape",   1, 8, 64, 14, 14, 0.125, "THW", False, True, 16),
    ("thw_l4_shape",   1, 8, 128, 7, 7, 0.125, "THW", False, True, 32),
]

# name -> (N, T, Cin, planes, H, W, stride); MVF(alpha=0.125, THW) on conv1
BLOCK_CASES = {
    "l3_like": (2, 4, 64, 16, 8, 8, 1),
    "l3_first": (1, 4, 32, 16, 10, 10, 2),
}
