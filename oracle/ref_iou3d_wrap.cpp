// TEST INFRASTRUCTURE (oracle): C-ABI doorway into the reference's own CPU implementation of the rotated BEV IoU,
// /root/reference/pcdet/ops/iou3d_nms/src/iou3d_cpu.cpp (boxes_iou_bev_cpu, :222-251), compiled from where it lies by
// oracle/build_ref.py into oracle/_ref/libiou3d_ref.so.  Nothing under virconv_amd/ may load this library.
#include <torch/torch.h>

int boxes_iou_bev_cpu(at::Tensor boxes_a_tensor, at::Tensor boxes_b_tensor, at::Tensor ans_iou_tensor);  // the reference's symbol

extern "C" int ref_boxes_iou_bev_cpu(const float* boxes_a, int n_a, const float* boxes_b, int n_b, float* out) {
  auto opt = torch::TensorOptions().dtype(torch::kFloat32);
  at::Tensor a = torch::from_blob(const_cast<float*>(boxes_a), {n_a, 7}, opt);
  at::Tensor b = torch::from_blob(const_cast<float*>(boxes_b), {n_b, 7}, opt);
  at::Tensor o = torch::from_blob(out, {n_a, n_b}, opt);
  return boxes_iou_bev_cpu(a, b, o);
}
