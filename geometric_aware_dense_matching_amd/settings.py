"""Runtime A/B switches of the package, in ONE place.  Every switch selects between a hand-written HIP path and the plain
form it replaces (the form the parity tests tie it to); all default to the HIP path (USE_SIDE_STREAMS and
STATIC_MATCH_ROWS are the two that default to off).  They are read at CALL time
(`settings.USE_X`), so a test or a benchmark may flip one between two calls; the `GDM_*` environment variables only set
the initial values.

    switch                     env                        off =
    USE_MFMA_CONV              GDM_MFMA_CONV              trunk 3x3 convolutions (eval) on MIOpen
    USE_MFMA_CONV_TRAIN        GDM_MFMA_CONV_TRAIN        trunk 3x3 convolutions' training forward / dgrad on MIOpen
    USE_MFMA_GEMM              GDM_MFMA_GEMM              large 1x1 mixes on hipBLASLt fp32
    USE_FUSED_UPCONV           GDM_FUSED_UPCONV           PSPUpsample(64,64) as low-resolution GEMM + gather (two kernels)
    USE_LOWRES_UPCONV_TRAIN    GDM_LOWRES_UPCONV_TRAIN    PSPUpsample training path = upsample + MIOpen convolution
    USE_SPLIT_PSP_TRAIN        GDM_SPLIT_PSP_TRAIN        PSPModule training path = upsampled priors + concat + 5F bottleneck
    USE_FUSED_LFA              GDM_FUSED_LFA              RandLA attentive pooling on the separate gather / GEMM / pooling kernels
    USE_GROUPED_SPLINE         GDM_GROUPED_SPLINE         128-channel SplineConv layers on the dense [M, 125*out] GEMM
    USE_FUSED_BN_TRAIN         GDM_FUSED_BN_TRAIN         training BatchNorm + activation on the torch modules
    USE_FUSED_SYNCBN           GDM_FUSED_SYNCBN           nn.SyncBatchNorm on torch's implementation
    USE_FUSED_MATCH_LOSS       GDM_FUSED_MATCH_LOSS       training similarity materialised by hipBLASLt, rows kernel for the circle loss
    USE_SIDE_STREAMS           GDM_SIDE_STREAMS=1         (default OFF: everything on the caller's stream) inference forks the mesh branch, the
                                                          neighbour pyramid and the point branch of each encoder stage onto side streams
                                                          (worth ~3 % of the step; see DESIGN.md "Side streams")
    UPCONV_MIN_CIN             GDM_UPCONV_MIN_CIN         (int) smallest Cin for the low-resolution form of conv3x3(upsample(x))
    USE_SPARSE_FINAL           GDM_SPARSE_FINAL           the last image stage (up_3 + final) on the full 2x map, then the gather with `choose`
                                                          (default: evaluated at the chosen pixels only -- inference, 1/32 of the pixels)
    USE_FUSED_HEADS            GDM_FUSED_HEADS            the nine per-point 1x1 convolutions after the embedding as library GEMMs + BN kernels
    USE_MFMA_WGRAD             GDM_MFMA_WGRAD             weight gradient of the trunk's 3x3 convolutions at Cin = 256 / 512 on MIOpen (fp32 implicit GEMM)
    USE_GATHERED_FINAL         GDM_GATHERED_FINAL         training: FinalStage (1x1 conv + LogSoftmax) on the N chosen pixels    FinalStage on all H*W pixels, then the gather
    USE_DIRECT_WGRAD           GDM_DIRECT_WGRAD           training: weight (+ bias) gradient of the small-channel 1x1 layers in one pass over fp32 rows        batched fp32 GEMM + sums
    USE_FUSED_ADAM             GDM_FUSED_ADAM             training: torch.optim.Adam(fused=True), one multi-tensor kernel                                        foreach Adam (~25 launches)
    USE_MFMA_GEMM_TRAIN        GDM_MFMA_GEMM_TRAIN        training: the large 1x1 products (PSPUpsample tap GEMMs, PSP bottleneck, 512 / 1024-channel fusion layers)
                                                          forward / input gradient / weight gradient on hipBLASLt fp32 batched GEMMs
    USE_GEMM_CONV1X1_TRAIN     GDM_GEMM_CONV1X1_TRAIN     training 1x1 convolutions through torch's convolution (MIOpen wgrad / bwd-data + NHWC transposes)
    USE_OWN_STEM               GDM_OWN_STEM               the stem (conv 7x7/2 + BN + ReLU + max-pool) as an MIOpen convolution + one fused BN/ReLU/pool launch
    USE_POINTWISE              GDM_POINTWISE              per-point 1x1 layers (point branch, fusion, decoder) as library GEMM + BN/activation kernel + torch.cat
    USE_MFMA_STRIDED           GDM_MFMA_STRIDED           stride-2 residual block (layer2.0) and the 1x1 downsample branches on MIOpen / hipBLASLt
    USE_POINTWISE_TRAIN        GDM_POINTWISE_TRAIN        training: forward / input gradient of the small 1x1 layers as hipBLASLt fp32 batched GEMMs
    USE_POINT_CHAIN            GDM_POINT_CHAIN            the RandLA stem layer and the first block's mlp1 as two launches of the per-point kernel
    USE_PSP_JOBS               GDM_PSP_JOBS               the four prior products of the pyramid-pooling module as four launches of the per-point kernel
    STATIC_MATCH_ROWS          GDM_STATIC_MATCH_ROWS=1    (default off) matching loss over all B*N rows with zero weight on the unselected
                                                          ones instead of compacting them: no host read; always on under graph capture
"""
import os


def _flag(name, default="1"):
    return os.environ.get(name, default) != "0"


USE_MFMA_CONV = _flag("GDM_MFMA_CONV")
USE_MFMA_CONV_TRAIN = _flag("GDM_MFMA_CONV_TRAIN")
USE_MFMA_GEMM = _flag("GDM_MFMA_GEMM")
USE_FUSED_UPCONV = _flag("GDM_FUSED_UPCONV")
USE_LOWRES_UPCONV_TRAIN = _flag("GDM_LOWRES_UPCONV_TRAIN")
USE_SPLIT_PSP_TRAIN = _flag("GDM_SPLIT_PSP_TRAIN")
USE_FUSED_LFA = _flag("GDM_FUSED_LFA")
USE_GROUPED_SPLINE = _flag("GDM_GROUPED_SPLINE")
USE_FUSED_BN_TRAIN = _flag("GDM_FUSED_BN_TRAIN")
USE_FUSED_SYNCBN = _flag("GDM_FUSED_SYNCBN")
USE_FUSED_MATCH_LOSS = _flag("GDM_FUSED_MATCH_LOSS")
USE_SIDE_STREAMS = _flag("GDM_SIDE_STREAMS", "0")
USE_SPARSE_FINAL = _flag("GDM_SPARSE_FINAL")
USE_FUSED_HEADS = _flag("GDM_FUSED_HEADS")
USE_MFMA_STRIDED = _flag("GDM_MFMA_STRIDED")
USE_POINTWISE = _flag("GDM_POINTWISE")
USE_OWN_STEM = _flag("GDM_OWN_STEM")
USE_GEMM_CONV1X1_TRAIN = _flag("GDM_GEMM_CONV1X1_TRAIN")
USE_MFMA_WGRAD = _flag("GDM_MFMA_WGRAD")
USE_MFMA_GEMM_TRAIN = _flag("GDM_MFMA_GEMM_TRAIN")
USE_GATHERED_FINAL = _flag("GDM_GATHERED_FINAL")
USE_DIRECT_WGRAD = _flag("GDM_DIRECT_WGRAD")
USE_FUSED_ADAM = _flag("GDM_FUSED_ADAM")
USE_POINTWISE_TRAIN = _flag("GDM_POINTWISE_TRAIN")       # training: forward / input gradient of the small 1x1 layers on the own fp32-MFMA per-point kernel (off: hipBLASLt batched GEMMs)
USE_POINT_CHAIN = _flag("GDM_POINT_CHAIN")               # the RandLA stem fc0 and the first block's mlp1 as one launch (gdm_pointwise_chain2_hip)
USE_PSP_JOBS = _flag("GDM_PSP_JOBS")                     # the four prior products of the pyramid-pooling module in one launch (gdm_pointwise_jobs_hip)
USE_PACKED_PRODUCERS = _flag("GDM_PACKED_PRODUCERS")     # psp_combine / the up-conv gather write the next GEMM's packed operand themselves
USE_TWO_STREAM_PIPELINE = _flag("GDM_TWO_STREAM_PIPELINE")     # with USE_SIDE_STREAMS: image / point streams run ahead of each other, one event per stage and direction
MESH_FORK_LATE = os.environ.get("GDM_MESH_FORK_LATE", "1") != "0"       # with USE_SIDE_STREAMS: the mesh fork is enqueued behind the embedding
PACK_MESH_ROWS = os.environ.get("GDM_PACK_MESH_ROWS", "1") != "0"         # with USE_SIDE_STREAMS: the model descriptors' matching rows are packed inside the mesh fork
SIDE_PARTS = os.environ.get("GDM_SIDE_PARTS", "mesh,point,pyr").split(",")     # development: which branches are forked
STATIC_MATCH_ROWS = _flag("GDM_STATIC_MATCH_ROWS", "0")
UPCONV_MIN_CIN = int(os.environ.get("GDM_UPCONV_MIN_CIN", "0"))
# ranks a SyncBatchNorm's group must span before its statistics are all-reduced (2: a group of one is plain BatchNorm).  1 sends a single
# rank through the collective as well -- the RCCL rehearsal on a one-GPU box (tests/test_gpu_train.py)
SYNCBN_MIN_WORLD = 2

ALL_SWITCHES = ("USE_MFMA_CONV", "USE_MFMA_CONV_TRAIN", "USE_MFMA_GEMM", "USE_FUSED_UPCONV", "USE_LOWRES_UPCONV_TRAIN",
                "USE_SPLIT_PSP_TRAIN", "USE_FUSED_LFA", "USE_GROUPED_SPLINE", "USE_FUSED_BN_TRAIN", "USE_FUSED_SYNCBN",
                "USE_FUSED_MATCH_LOSS", "USE_SIDE_STREAMS", "USE_SPARSE_FINAL", "USE_FUSED_HEADS", "USE_MFMA_STRIDED", "USE_POINTWISE", "USE_OWN_STEM", "USE_GEMM_CONV1X1_TRAIN", "USE_MFMA_WGRAD", "USE_MFMA_GEMM_TRAIN", "USE_GATHERED_FINAL", "USE_DIRECT_WGRAD", "USE_PSP_JOBS", "USE_POINT_CHAIN", "USE_POINTWISE_TRAIN")
# the switches behind which split-bf16 (x3, fp32 accumulate) products run in the eval forward: all off = fp32 products everywhere
# (hipBLASLt / MIOpen / fp32 FMA kernels); the matching kernel's precision is its own argument (matching.match_frames(precision=))
SPLIT_BF16_SWITCHES = ("USE_MFMA_CONV", "USE_MFMA_GEMM", "USE_FUSED_UPCONV", "USE_SPARSE_FINAL", "USE_FUSED_HEADS", "USE_MFMA_STRIDED",
                       "USE_OWN_STEM")
