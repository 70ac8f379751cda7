// binding.cpp -- the boundary the reference actually binds: a pybind11 torch extension whose functions take torch::Tensor
// (kernels/setup.py:157-245 builds seven such modules under the package name `qserve_backend`).  Every function below
// keeps the reference's name, argument order, in-place / returned outputs and TORCH_CHECK behaviour and reduces to ONE call
// into the C ABI of libqserve_amd.so (include/qserve_amd.h) on the current HIP stream - what a maintainer of the reference
// would put in place of the CUDA bodies.  One shared object carries the seven modules as sub-modules; the Python package
// qserve_backend_ext registers them under the reference's import names.
//
//   qgemm_w4a8_per_chn.gemm_forward_cuda ........ kernels/csrc/qgemm/w4a8_per_chn/pybind.cpp:13-16, gemm_cuda.cu:596-652
//   qgemm_w4a8_per_group.gemm_forward_cuda ...... kernels/csrc/qgemm/w4a8_per_group/pybind.cpp:13-16, gemm_cuda.cu:630-702
//   qgemm_w8a8.w8a8_gemm_forward_cuda ........... kernels/csrc/qgemm/w8a8/pybind.cpp:13-16
//   fused_attention.{single_query_attention, apply_bias_rope_update_kv_cache, compute_padding_offsets}
//                                                 kernels/csrc/fused_attention/fused_attention.cpp:150-256
//   fused_kernels.{invoke_quant, invoke_quant_fuse_sum} (per-token overloads) ....... kernels/csrc/fused.cpp:47-71
//   layernorm_ops.{rms_norm, rms_norm_general, rms_norm_general_fuse_sum} .......... kernels/csrc/layernorm.cpp:47-72
//   activation_ops.silu_and_mul .............................................. kernels/csrc/activation.cpp:25-39
// The W8A8-only functions (invoke_dequant*, gelu_*, per-tensor-scale overloads) are out of scope:
// they exist as names and raise.
#include <torch/extension.h>
#include <ATen/hip/HIPContext.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

#include "qserve_amd.h"

namespace {

// (PyTorch-ROCm presents HIP devices under the device type "cuda": the stream and guard types are the "masquerading" ones)
void* cur_stream() { return reinterpret_cast<void*>(c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream()); }
using DeviceGuard = c10::hip::OptionalHIPGuardMasqueradingAsCUDA;

#define QS_CALL(expr)                                  \
    do {                                               \
        const int rc_ = (expr);                        \
        TORCH_CHECK(rc_ == 0, qs_last_error(), " (code ", rc_, ")"); \
    } while (0)

void need(const torch::Tensor& t, at::ScalarType dt, const char* name, bool contiguous = true) {
    TORCH_CHECK(t.is_cuda(), name, " must be on CUDA");
    TORCH_CHECK(t.scalar_type() == dt, "expected scalar type ", dt, " for ", name, " but found ", t.scalar_type());
    if (contiguous) TORCH_CHECK(t.is_contiguous(), name, " must be contiguous");
}

// ---- qgemm_w4a8_per_chn / per_group / w8a8 -----------------------------------------------------------------------------
// M, N come from out_feats and K from in_feats (as the reference derives them, gemm_cuda.cu:604-613); everything else must fit,
// or the kernels would read out of bounds.  pack = weights per byte of `kernel` (2 for the 4-bit layouts).
void gemm_shapes(const torch::Tensor& in_feats, const torch::Tensor& kernel, const torch::Tensor& out_feats, int pack) {
    TORCH_CHECK(in_feats.dim() >= 2 && kernel.dim() == 2 && out_feats.dim() >= 2, "in_feats / kernel / out_feats must be matrices");
    const int64_t M = out_feats.size(-2), N = out_feats.size(-1), K = in_feats.size(1);
    TORCH_CHECK(in_feats.numel() == M * K, "in_feats holds ", in_feats.numel(), " values, out_feats implies ", M, " x ", K);
    TORCH_CHECK(kernel.size(0) == N && kernel.size(1) * pack == K, "kernel ", kernel.sizes(), " does not match N = ", N, ", K = ", K);
}

void gemm_per_chn(torch::Tensor in_feats, torch::Tensor kernel, torch::Tensor wscales, torch::Tensor ascales,
                  torch::Tensor w_szs, torch::Tensor a_ssums, torch::Tensor out_feats) {
    need(in_feats, at::kChar, "in_feats"); need(kernel, at::kChar, "kernel"); need(out_feats, at::kHalf, "out_feats");
    need(wscales, at::kHalf, "wscales"); need(ascales, at::kHalf, "ascales"); need(w_szs, at::kHalf, "w_szs");
    need(a_ssums, at::kHalf, "a_ssums");
    gemm_shapes(in_feats, kernel, out_feats, 2);
    TORCH_CHECK(wscales.numel() >= out_feats.size(-1) && w_szs.numel() >= out_feats.size(-1), "wscales / w_szs must hold one value per output channel");
    TORCH_CHECK(ascales.numel() >= out_feats.size(-2) && a_ssums.numel() >= out_feats.size(-2), "ascales / a_ssums must hold one value per token");
    const DeviceGuard guard(in_feats.device());
    // shapes as the reference takes them (gemm_cuda.cu:604-613)
    QS_CALL(qs_w4a8_per_chn_gemm(in_feats.data_ptr<int8_t>(), kernel.data_ptr<int8_t>(), wscales.data_ptr(), ascales.data_ptr(),
                                 w_szs.data_ptr(), a_ssums.data_ptr(), out_feats.data_ptr(), (int)out_feats.size(-2),
                                 (int)out_feats.size(-1), (int)in_feats.size(1), cur_stream()));
}
void gemm_per_group(torch::Tensor in_feats, torch::Tensor kernel, torch::Tensor zeros, torch::Tensor scales_i8,
                    torch::Tensor wscales, torch::Tensor ascales, torch::Tensor out_feats) {
    need(in_feats, at::kChar, "in_feats"); need(kernel, at::kChar, "kernel"); need(zeros, at::kChar, "zeros");
    need(scales_i8, at::kChar, "scales_i8"); need(wscales, at::kHalf, "wscales"); need(ascales, at::kHalf, "ascales");
    need(out_feats, at::kHalf, "out_feats");
    gemm_shapes(in_feats, kernel, out_feats, 2);
    {
        const int64_t N = out_feats.size(-1), K = in_feats.size(1);
        TORCH_CHECK(zeros.numel() >= (K / 128) * N && scales_i8.numel() >= (K / 128) * N, "zeros / scales_i8 must hold K/128 x N values");
        TORCH_CHECK(wscales.numel() >= N && ascales.numel() >= out_feats.size(-2), "wscales / ascales: one value per channel / token");
    }
    const DeviceGuard guard(in_feats.device());
    QS_CALL(qs_w4a8_per_group_gemm(in_feats.data_ptr<int8_t>(), kernel.data_ptr<int8_t>(), zeros.data_ptr<int8_t>(),
                                   scales_i8.data_ptr<int8_t>(), wscales.data_ptr(), ascales.data_ptr(), out_feats.data_ptr(),
                                   (int)out_feats.size(-2), (int)out_feats.size(-1), (int)in_feats.size(1), cur_stream()));
}
void gemm_w8a8(torch::Tensor in_feats, torch::Tensor kernel, torch::Tensor wscales, torch::Tensor ascales,
               torch::Tensor out_feats) {
    need(in_feats, at::kChar, "in_feats"); need(kernel, at::kChar, "kernel"); need(wscales, at::kHalf, "wscales");
    need(ascales, at::kHalf, "ascales"); need(out_feats, at::kHalf, "out_feats");
    gemm_shapes(in_feats, kernel, out_feats, 1);
    const DeviceGuard guard(in_feats.device());
    QS_CALL(qs_w8a8_gemm(in_feats.data_ptr<int8_t>(), kernel.data_ptr<int8_t>(), wscales.data_ptr(), ascales.data_ptr(),
                         out_feats.data_ptr(), (int)out_feats.size(-2), (int)out_feats.size(-1), (int)in_feats.size(1),
                         cur_stream()));
}

// ---- fused_attention ---------------------------------------------------------------------------------------------------
torch::Tensor single_query_attention(const torch::Tensor q, const torch::Tensor k, const torch::Tensor v,
                                     const torch::Tensor kv_pointers, c10::optional<const torch::Tensor> length_per_sample_,
                                     c10::optional<const torch::Tensor> alibi_slopes_, const int memory_max_seqlen,
                                     const int tokens_per_block, const int size_per_token, const int timestep,
                                     const int rotary_embedding_dim, const float rotary_base, const bool neox_rotary_style,
                                     const bool int4_kv_cache, const bool kv_cache_with_zeros) {
    need(q, at::kHalf, "q", false); need(k, at::kHalf, "k", false); need(v, at::kHalf, "v", false);
    need(kv_pointers, at::kLong, "kv_pointers");
    const int64_t batch = kv_pointers.size(0), nheads = q.size(1), nheads_kv = k.size(1), headdim = k.size(-1);
    // fused_attention.cpp:179-180
    TORCH_CHECK(k.stride(2) == 1 && k.stride(1) == headdim && v.stride(2) == 1 && v.stride(1) == headdim,
                "k and v must have stride(2) == 1 and stride(1) == head_dim");
    TORCH_CHECK(q.stride(2) == 1 && q.stride(1) == headdim, "q must have stride(2) == 1 and stride(1) == head_dim");
    const int32_t* lens = nullptr;
    if (length_per_sample_.has_value()) {
        const auto& l = length_per_sample_.value();
        need(l, at::kInt, "length_per_sample");
        TORCH_CHECK(l.dim() == 1 && l.size(0) == batch, "length_per_sample must have shape (batch_size)");
        lens = l.data_ptr<int32_t>();
    }
    if (alibi_slopes_.has_value()) {
        // checked, then ignored - the reference's behaviour (fused_attention.cpp:193-199 checks, :91 never stores the pointer,
        // decoderMaskedMultiheadAttentionTemplate.hpp:1604-1615 is commented out)
        const auto& a = alibi_slopes_.value();
        need(a, at::kFloat, "alibi_slopes");
        TORCH_CHECK(a.dim() == 1 && a.size(0) == nheads, "alibi_slopes must have shape (nheads)");
    }
    const DeviceGuard guard(q.device());       // fused_attention.cpp:203
    torch::Tensor out = torch::empty({q.size(0), nheads, headdim}, q.options());
    QS_CALL(qs_single_query_attention(q.data_ptr(), k.data_ptr(), v.data_ptr(), kv_pointers.data_ptr<int64_t>(), lens,
                                      out.data_ptr(), (int)batch, (int)nheads, (int)nheads_kv, (int)headdim, q.stride(0),
                                      k.stride(0), (int)kv_pointers.size(-1), memory_max_seqlen, tokens_per_block,
                                      size_per_token, timestep, rotary_embedding_dim, rotary_base, neox_rotary_style,
                                      int4_kv_cache, kv_cache_with_zeros, cur_stream()));
    return out;
}
void apply_bias_rope_update_kv_cache(torch::Tensor qkv, const torch::Tensor seq_lens, const torch::Tensor padding_offset,
                                     c10::optional<const torch::Tensor> kv_pointers_, const int head_num,
                                     const int kv_head_num, const int seq_len, const int tokens_per_block,
                                     const int size_per_token, const int rotary_embedding_dim,
                                     const float rotary_embedding_base, const int rotary_embedding_max_positions,
                                     const bool neox_rotary_style, const bool int4_kv_cache, const bool kv_cache_with_zeros) {
    need(qkv, at::kHalf, "qkv"); need(seq_lens, at::kInt, "seq_lens"); need(padding_offset, at::kInt, "padding_offset");
    const int64_t* kvp = nullptr;
    int mb = 0;
    if (kv_pointers_.has_value()) {
        need(kv_pointers_.value(), at::kLong, "kv_pointers");
        kvp = kv_pointers_.value().data_ptr<int64_t>();
        mb = (int)kv_pointers_.value().size(-1);
    }
    const DeviceGuard guard(qkv.device());
    QS_CALL(qs_apply_bias_rope_update_kv_cache(qkv.data_ptr(), seq_lens.data_ptr<int32_t>(), padding_offset.data_ptr<int32_t>(),
                                               kvp, (int)qkv.size(0), (int)seq_lens.size(0), mb, head_num, kv_head_num, seq_len,
                                               tokens_per_block, size_per_token, rotary_embedding_dim, rotary_embedding_base,
                                               rotary_embedding_max_positions, neox_rotary_style, int4_kv_cache,
                                               kv_cache_with_zeros, cur_stream()));
}
torch::Tensor compute_padding_offsets(const torch::Tensor cu_seqlens, const int max_seqlen, const int tot_num_tokens) {
    need(cu_seqlens, at::kInt, "cu_seqlens");
    const DeviceGuard guard(cu_seqlens.device());
    torch::Tensor out = torch::empty({tot_num_tokens}, cu_seqlens.options());
    QS_CALL(qs_compute_padding_offsets(out.data_ptr<int32_t>(), cu_seqlens.data_ptr<int32_t>(), (int)cu_seqlens.size(0) - 1,
                                       max_seqlen, cur_stream()));
    return out;
}

// ---- fused_kernels (per-token overloads) ---------------------------------------------------------------------------------
// out and input hold the same number of values; the per-token vectors one value per row
void row_shapes(const torch::Tensor& out, const torch::Tensor& input, int hidden, const torch::Tensor* scale,
                const torch::Tensor* sum) {
    TORCH_CHECK(hidden > 0 && out.numel() == input.numel(), "out must hold as many values as input");
    const int64_t rows = input.numel() / hidden;
    if (scale) TORCH_CHECK(scale->numel() >= rows, "scale must hold one value per token");
    if (sum) TORCH_CHECK(sum->numel() >= rows, "input_sum must hold one value per token");
}

void invoke_quant(torch::Tensor& out, torch::Tensor& input, torch::Tensor& scale) {
    need(out, at::kChar, "out"); need(input, at::kHalf, "input"); need(scale, at::kHalf, "scale");
    const int hidden = (int)input.size(-1);
    row_shapes(out, input, hidden, &scale, nullptr);
    const DeviceGuard guard(out.device());
    QS_CALL(qs_invoke_quant(out.data_ptr<int8_t>(), input.data_ptr(), nullptr, scale.data_ptr(), (int)(input.numel() / hidden),
                            hidden, cur_stream()));
}
void invoke_quant_fuse_sum(torch::Tensor& out, torch::Tensor& input, torch::Tensor& input_sum, torch::Tensor& scale) {
    need(out, at::kChar, "out"); need(input, at::kHalf, "input"); need(input_sum, at::kHalf, "input_sum");
    need(scale, at::kHalf, "scale");
    const int hidden = (int)input.size(-1);
    row_shapes(out, input, hidden, &scale, &input_sum);
    const DeviceGuard guard(out.device());
    QS_CALL(qs_invoke_quant(out.data_ptr<int8_t>(), input.data_ptr(), input_sum.data_ptr(), scale.data_ptr(),
                            (int)(input.numel() / hidden), hidden, cur_stream()));
}

// ---- layernorm_ops -------------------------------------------------------------------------------------------------------
void rms_norm(torch::Tensor& out, torch::Tensor& input, torch::Tensor& weight, float epsilon, bool use_quant) {
    TORCH_CHECK(!use_quant, "rms_norm(use_quant=True) belongs to the W8A8 path (out of scope)");
    need(out, at::kHalf, "out"); need(input, at::kHalf, "input"); need(weight, at::kHalf, "weight");
    const int hidden = (int)input.size(-1);
    row_shapes(out, input, hidden, nullptr, nullptr);
    TORCH_CHECK(weight.numel() == hidden, "weight must hold one value per hidden element");
    const DeviceGuard guard(out.device());
    QS_CALL(qs_rms_norm(out.data_ptr(), input.data_ptr(), weight.data_ptr(), epsilon, (int)(input.numel() / hidden), hidden,
                        cur_stream()));
}
void rms_norm_general(torch::Tensor& out, torch::Tensor& input, torch::Tensor& weight, torch::Tensor& scaling, float epsilon,
                      bool use_per_token_quant) {
    TORCH_CHECK(use_per_token_quant, "rms_norm_general per-tensor scaling belongs to the W8A8 path (out of scope)");
    need(out, at::kChar, "out"); need(input, at::kHalf, "input"); need(weight, at::kHalf, "weight");
    need(scaling, at::kHalf, "scaling");
    const int hidden = (int)input.size(-1);
    row_shapes(out, input, hidden, &scaling, nullptr);
    TORCH_CHECK(weight.numel() == hidden, "weight must hold one value per hidden element");
    const DeviceGuard guard(out.device());
    QS_CALL(qs_rms_norm_general(out.data_ptr<int8_t>(), input.data_ptr(), weight.data_ptr(), nullptr, scaling.data_ptr(), epsilon,
                                (int)(input.numel() / hidden), hidden, cur_stream()));
}
void rms_norm_general_fuse_sum(torch::Tensor& out, torch::Tensor& input, torch::Tensor& weight, torch::Tensor& input_sum,
                               torch::Tensor& scaling, float epsilon, bool use_per_token_quant) {
    TORCH_CHECK(use_per_token_quant, "rms_norm_general_fuse_sum has no per-tensor variant (the reference asserts)");
    need(out, at::kChar, "out"); need(input, at::kHalf, "input"); need(weight, at::kHalf, "weight");
    need(input_sum, at::kHalf, "input_sum"); need(scaling, at::kHalf, "scaling");
    const int hidden = (int)input.size(-1);
    row_shapes(out, input, hidden, &scaling, &input_sum);
    TORCH_CHECK(weight.numel() == hidden, "weight must hold one value per hidden element");
    const DeviceGuard guard(out.device());
    QS_CALL(qs_rms_norm_general(out.data_ptr<int8_t>(), input.data_ptr(), weight.data_ptr(), input_sum.data_ptr(),
                                scaling.data_ptr(), epsilon, (int)(input.numel() / hidden), hidden, cur_stream()));
}

// ---- activation_ops ------------------------------------------------------------------------------------------------------
void silu_and_mul(torch::Tensor& out, torch::Tensor& input) {
    need(out, at::kHalf, "out"); need(input, at::kHalf, "input");
    const int d = (int)input.size(-1) / 2;
    TORCH_CHECK(input.size(-1) % 2 == 0 && out.numel() * 2 == input.numel(), "out must hold half as many values as input");
    const DeviceGuard guard(out.device());
    QS_CALL(qs_silu_and_mul(out.data_ptr(), input.data_ptr(), (int)(input.numel() / input.size(-1)), d, cur_stream()));
}

void out_of_scope(py::args, py::kwargs) {
    TORCH_CHECK(false, "this op belongs to the W8A8 / GELU paths of the reference, which are out of scope for the W4A8KV4 hot "
                       "path (SURVEY.md 2 rows 6-9)");
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "qserve_backend as a compiled torch extension over libqserve_amd.so (MI355X / gfx950)";
    auto chn = m.def_submodule("qgemm_w4a8_per_chn");
    chn.def("gemm_forward_cuda", &gemm_per_chn, "W4A8 per-channel GEMM", py::arg("in_feats"), py::arg("kernel"),
            py::arg("wscales"), py::arg("ascales"), py::arg("w_szs"), py::arg("a_ssums"), py::arg("out_feats"));
    auto grp = m.def_submodule("qgemm_w4a8_per_group");
    grp.def("gemm_forward_cuda", &gemm_per_group, "W4A8 per-group (g128) GEMM", py::arg("in_feats"), py::arg("kernel"),
            py::arg("zeros"), py::arg("scales_i8"), py::arg("wscales"), py::arg("ascales"), py::arg("out_feats"));
    auto w8 = m.def_submodule("qgemm_w8a8");
    w8.def("w8a8_gemm_forward_cuda", &gemm_w8a8, "W8A8 GEMM", py::arg("in_feats"), py::arg("kernel"), py::arg("wscales"),
           py::arg("ascales"), py::arg("out_feats"));
    auto att = m.def_submodule("fused_attention");
    att.def("single_query_attention", &single_query_attention, "decode attention over the paged quantised KV cache",
            py::arg("q"), py::arg("k"), py::arg("v"), py::arg("kv_pointers"), py::arg("length_per_sample"),
            py::arg("alibi_slopes"), py::arg("memory_max_seqlen"), py::arg("tokens_per_block"), py::arg("size_per_token"),
            py::arg("timestep"), py::arg("rotary_embedding_dim"), py::arg("rotary_base"), py::arg("neox_rotary_style"),
            py::arg("int4_kv_cache"), py::arg("kv_cache_with_zeros"));
    att.def("apply_bias_rope_update_kv_cache", &apply_bias_rope_update_kv_cache,
            "(context stage) add bias, apply rope and update kv cache", py::arg("qkv"), py::arg("seq_lens"),
            py::arg("padding_offset"), py::arg("kv_pointers"), py::arg("head_num"), py::arg("kv_head_num"), py::arg("seq_len"),
            py::arg("tokens_per_block"), py::arg("size_per_token"), py::arg("rotary_embedding_dim"),
            py::arg("rotary_embedding_base"), py::arg("rotary_embedding_max_positions"), py::arg("neox_rotary_style"),
            py::arg("int4_kv_cache"), py::arg("kv_cache_with_zeros"));
    att.def("compute_padding_offsets", &compute_padding_offsets, "compute padding offsets", py::arg("cu_seqlens"),
            py::arg("max_seqlen"), py::arg("tot_num_tokens"));
    auto fk = m.def_submodule("fused_kernels");
    fk.def("invoke_quant", &invoke_quant, "Quant.", py::arg("out"), py::arg("input"), py::arg("scale"));
    fk.def("invoke_quant_fuse_sum", &invoke_quant_fuse_sum, "Quant & get input sum.", py::arg("out"), py::arg("input"),
           py::arg("input_sum"), py::arg("scale"));
    fk.def("invoke_dequant", &out_of_scope);
    fk.def("invoke_dequant_add_residual", &out_of_scope);
    auto ln = m.def_submodule("layernorm_ops");
    ln.def("rms_norm", &rms_norm, py::arg("out"), py::arg("input"), py::arg("weight"), py::arg("epsilon"),
           py::arg("use_quant") = false);
    ln.def("rms_norm_general", &rms_norm_general, py::arg("out"), py::arg("input"), py::arg("weight"), py::arg("scaling"),
           py::arg("epsilon"), py::arg("use_per_token_quant") = false);
    ln.def("rms_norm_general_fuse_sum", &rms_norm_general_fuse_sum, py::arg("out"), py::arg("input"), py::arg("weight"),
           py::arg("input_sum"), py::arg("scaling"), py::arg("epsilon"), py::arg("use_per_token_quant") = false);
    ln.def("invoke_dequant_add_residual_rms_norm_quant", &out_of_scope);
    auto act = m.def_submodule("activation_ops");
    act.def("silu_and_mul", &silu_and_mul, "Activation function used in SwiGLU.", py::arg("out"), py::arg("input"));
    act.def("gelu_new", &out_of_scope);
    act.def("gelu_fast", &out_of_scope);
    act.def("invoke_dequant_silu_and_mul_quant", &out_of_scope);
}
