"""Kernel name -> bench.py profiling class (the ProfScope class the kernel is launched under, csrc/*.hip; include/ssv_hip.h SSV_PROF_*), shared by the
rocprofv3 aggregators (pmc_traffic.py, pmc_mfma.py, kstats_steady.py): counters and HIP-event times must be split the same way or a class's
bytes / busy cycles get divided by another class's time.  First match wins; tests/test_host_cpu.py checks the table against the sources."""

# (class, substring of the kernel name).  Winograd transforms belong to the product they serve: input / plain or statistics output -> the forward-kernel
# family (the transformed-domain GEMMs ARE conv_fwd_k launches), gated output (<2>, <3>: a data gradient's epilogue) -> conv_dgrad, dY / filter-gradient
# transforms -> weight gradient.  The stem's rows-in-LDS kernels (round 4) are the stem's forward / weight gradient.
CLASSES = (
    ("conv_fwd", "conv_fwd_k"), ("conv_fwd", "stem_fwd_rows_k"),
    ("conv_dgrad", "conv_dgrad_k"), ("conv_dgrad", "wino_output_k<2"), ("conv_dgrad", "wino_output_k<3"), ("conv_dgrad", "wino44_output_k<2"), ("conv_dgrad", "wino44_output_k<3"),
    ("conv_wgrad", "conv_wgrad_k"), ("conv_wgrad", "stem_wgrad_rows_k"), ("conv_wgrad", "wgrad_reduce"),
    ("conv_wgrad", "wino_dy_k"), ("conv_wgrad", "wino_dfilter_k"), ("conv_wgrad", "wino44_dy_k"), ("conv_wgrad", "wino44_dy_both_k"), ("conv_wgrad", "wino44_dfilter_k"),
    ("conv_fwd", "wino_input_k"), ("conv_fwd", "wino_output_k"), ("conv_fwd", "wino44_input_k"), ("conv_fwd", "wino44_output_k"),
    ("misc", "wino_filter_k"), ("misc", "wino44_filter_k"),
    ("bn_fwd", "bn_stats"), ("bn_fwd", "bn_apply"), ("bn_fwd", "bn_partials_coarsen"), ("bn_fwd", "bn_relu_maxpool_fwd"),
    ("bn_bwd", "bn_bwd"), ("bn_bwd", "bn_pool_bwd"), ("bn_bwd", "bn_sums_coarsen"),
    ("attn", "attn_"), ("norm", "ln_"),
    ("loss", "ntxent_"), ("loss", "l2norm_"), ("loss", "mse_pair_k"), ("loss", "barlow_cgrad_k"), ("loss", "sum_partials_k"), ("loss", "dino_"), ("loss", "softmax_ce_k"),
    ("loss", "ce_reduce_k"), ("loss", "negdot_pair_k"), ("loss", "finish_sum_k"), ("loss", "relic_"), ("loss", "moco_"),
    ("optim", "sgd_"), ("optim", "adamw_k"), ("optim", "adamw_tick_k"), ("optim", "ema_k"), ("aug", "aug_"), ("aug", "multicrop"), ("aug", "center_view_k"),
    ("pool", "maxpool_"), ("pool", "gap_"),
    ("misc", "gelu_"), ("misc", "colsum_"), ("misc", "wn_fwd_k"), ("misc", "wn_bwd_k"), ("misc", "knn_agree_k"), ("misc", "knn_fused_k"), ("misc", "knn_finish_k"), ("misc", "zero_count_k"), ("misc", "scale_k"), ("misc", "add_k"),
    ("misc", "fill_k"), ("misc", "pad_channels_k"), ("misc", "group_expand_k"), ("misc", "group_extract_k"), ("misc", "filter_transpose_k"), ("misc", "nchw_to_nhwc_k"),
    ("misc", "nhwc_to_nchw_k"), ("misc", "queue_push_k"), ("misc", "queue_advance_k"), ("misc", "vit_embed_"), ("misc", "split_planes_k"),
)
CONV_FAMILY = ("conv_fwd", "conv_dgrad", "conv_wgrad")
# host functions whose kernels run under another class's scope than the table gives them - both inside the conv family, so the family sums agree:
# fc2's data gradient of the ViT FFN is a FORWARD-kernel launch (GELU-derivative epilogue) timed as a data gradient (DINO only)
SCOPE_EXCEPTIONS = {"ssv_linear_fwd_gelugrad"}


# template-argument count of the GEMM kernels whose LAST argument is SP (csrc/conv_mfma.hip): true = the launch multiplies in SSV_ARITH_BF16X3 (six bf16 piece
# products per fp32 product on v_mfma_f32_16x16x32_bf16), false = on v_mfma_f32_32x32x2_f32
_SP_ARGS = {"conv_fwd_k": 15, "conv_wgrad_k": 12, "conv_dgrad_k": 8}


def is_bf16x3(name):
    """Does this kernel (demangled name from a rocprofv3 trace) run its products as bf16 pieces?"""
    import re
    m = re.search(r"(conv_fwd_k|conv_wgrad_k|conv_dgrad_k)<([^>]*)>", name)
    if not m:
        return False
    args = [a.strip() for a in m.group(2).split(",")]
    return len(args) == _SP_ARGS[m.group(1)] and args[-1] in ("true", "1", "2")        # conv_fwd_k: 1 = one accumulator per tile, 2 = two (long contractions)


def classify(name):
    for cls, key in CLASSES:
        if key in name:
            return cls
    return "other"


def launches_by_scope(csrc_dir):
    """[(file, host function, [ProfScope classes named in it], [kernel names it launches directly])] parsed from the sources: the table above must
    agree with the scope every kernel is launched under."""
    import glob
    import os
    import re
    out = []
    for f in sorted(glob.glob(os.path.join(csrc_dir, "*.hip"))):
        text = open(f).read()
        for m in re.finditer(r'^(?:extern "C" |static |template <[^\n]*>\n)?[\w:<> \*]+\s+\**(\w+)\([^;{]*\)\s*\{\n(.*?)^\}', text, re.S | re.M):
            body = m.group(2)
            scopes = [c.lower() for c in re.findall(r"SSV_PROF_(\w+)", " ".join(re.findall(r"ProfScope ps\(([^;]+);", body)))]
            kernels = [k + (t or "") for k, t in re.findall(r"hipLaunchKernelGGL\(\(?\s*(\w+)(<[^>]*>)?", body)]
            if scopes and kernels:
                out.append((os.path.basename(f), m.group(1), scopes, kernels))
    return out


def in_conv_family(name):
    """Is this kernel part of the conv implicit-GEMM family bench.py's roofline prices (forward / data gradient / weight gradient kernels, their split-K
    reduces, the stem kernels, the Winograd transforms that run under a conv class)?"""
    return classify(name) in CONV_FAMILY
