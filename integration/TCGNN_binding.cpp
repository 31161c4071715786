// TCGNN_binding.cpp - the pybind11 torch extension INTEGRATION.md section B describes, complete and compiled.
//
// What a maintainer of the reference would keep of TCGNN_conv/TCGNN.cpp: the module (same seven names, TCGNN.cpp:260-272), the
// torch-typed wrappers with their CHECK_INPUT lines (TCGNN.cpp:54-56, :63-150, :172-256) - and nothing else.  The bodies that
// called the *_cuda launchers of TCGNN_kernel.cu (declared at TCGNN.cpp:13-52) call the C ABI of include/tcgnn.h instead;
// TCGNN_kernel.cu, nvcc and thrust are gone.  Built by integration/setup.py as a plain CppExtension (no .cu file, so torch's
// hipify step never runs) that links libtcgnn_hip.so.  tests/test_binding.py drives the three kernels through THIS module
// as a second backend next to the ctypes one (tc-gnn_atc23_amd/TCGNN.py).
#include <torch/extension.h>
#include <c10/hip/HIPStream.h>
#include <c10/core/DeviceGuard.h>
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdio>
#include <list>
#include <tuple>
#include <vector>

#include <tcgnn.h>

#define CHECK_CUDA(x) TORCH_CHECK(x.is_cuda(), #x " must be a CUDA tensor")            // TCGNN.cpp:54
#define CHECK_CONTIGUOUS(x) TORCH_CHECK(x.is_contiguous(), #x " must be contiguous")   // TCGNN.cpp:55
#define CHECK_INPUT(x) CHECK_CUDA(x); CHECK_CONTIGUOUS(x)                              // TCGNN.cpp:56

namespace {

void tcgnn_check(int st, const char* what) {
  TORCH_CHECK(st == TCGNN_OK, what, " failed: ", tcgnn_status_string(st), " (", tcgnn_last_error(), ")");
}

void* current_stream(const torch::Tensor& t) { return c10::hip::getCurrentHIPStream(t.get_device()).stream(); }

// the extension works on the device its tensors live on, not on whatever device happens to be current (plan allocations and
// kernel launches follow the current device) - what torch's own operators do with a device guard (r2 ADVICE)
using DeviceGuard = c10::DeviceGuard;   // (dispatches to the HIP guard registered for "cuda" tensors on ROCm)

// X must hold one row per node of the graph: a shorter matrix would be read out of bounds by the staging pass and the gathers
// (the reference indexes input by the same row ids, TCGNN_kernel.cu:417-427, without a check)
void check_rows(const torch::Tensor& input, const torch::Tensor& nodePointer) {
  TORCH_CHECK(input.dim() == 2, "input must be a 2-D [num_nodes, dim] matrix");
  TORCH_CHECK(nodePointer.numel() >= 1 && input.size(0) == nodePointer.numel() - 1, "input has ", input.size(0), " rows, the graph has ",
              nodePointer.numel() - 1, " nodes");
}

// One plan per graph.  The reference hands the same five tensors to every call (gnn_conv.py:31,56), so the packed tile stream
// is built at first sight and found again by (address, length, in-place version) of each; the entry keeps the tensors alive,
// so an address cannot be recycled under it.
struct PlanEntry {
  std::vector<std::tuple<const void*, int64_t, int64_t>> key;
  std::vector<torch::Tensor> keep;
  tcgnn_plan* plan = nullptr;
};
std::list<PlanEntry>& plan_cache() { static std::list<PlanEntry> c; return c; }
constexpr size_t kPlanCacheSize = 8;

tcgnn_plan* plan_for(const torch::Tensor& nodePointer, const torch::Tensor& edgeList, const torch::Tensor& blockPartition,
                     const torch::Tensor& edgeToColumn, const torch::Tensor& edgeToRow) {
  const torch::Tensor* ts[5] = {&nodePointer, &edgeList, &blockPartition, &edgeToColumn, &edgeToRow};
  std::vector<std::tuple<const void*, int64_t, int64_t>> key;
  for (auto* t : ts) key.emplace_back(t->data_ptr(), t->numel(), (int64_t)t->_version());
  auto& cache = plan_cache();
  for (auto it = cache.begin(); it != cache.end(); ++it)
    if (it->key == key) { cache.splice(cache.begin(), cache, it); return cache.front().plan; }
  for (auto* t : ts) TORCH_CHECK(t->scalar_type() == torch::kInt32, "expected scalar type Int");   // what data_ptr<int>() raises in the reference
  TORCH_CHECK(edgeToColumn.numel() >= edgeList.numel() && edgeToRow.numel() >= edgeList.numel(), "edgeToColumn / edgeToRow are shorter than edgeList");
  PlanEntry e;
  e.key = key;
  for (auto* t : ts) e.keep.push_back(*t);
  tcgnn_check(tcgnn_plan_create(nodePointer.data_ptr<int>(), edgeList.data_ptr<int>(), blockPartition.data_ptr<int>(),
                                edgeToColumn.data_ptr<int>(), edgeToRow.data_ptr<int>(), (int32_t)(nodePointer.size(0) - 1),
                                edgeList.size(0), (int32_t)blockPartition.size(0), current_stream(nodePointer), &e.plan),
              "tcgnn_plan_create");
  cache.push_front(std::move(e));
  while (cache.size() > kPlanCacheSize) {
    (void)hipDeviceSynchronize();   // kernels still reading the evicted plan - on ANY stream - finish first
    tcgnn_plan_destroy(cache.back().plan);
    cache.pop_back();
  }
  return cache.front().plan;
}

struct Workspace {
  torch::Tensor buf;
  char* ptr;
  size_t bytes;
  Workspace(tcgnn_plan* plan, int D, const torch::Tensor& like) {
    const size_t need = tcgnn_workspace_bytes(plan, D);
    buf = torch::empty({(int64_t)need + 256}, like.options().dtype(torch::kUInt8));   // torch's caching allocator: no hipMalloc per call
    ptr = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(buf.data_ptr()) + 255) & ~(uintptr_t)255);
    bytes = (size_t)buf.numel() - (size_t)(ptr - static_cast<char*>(buf.data_ptr()));
  }
};

}  // namespace

// TCGNN.cpp:63-86
std::vector<torch::Tensor> spmm_forward(torch::Tensor input, torch::Tensor nodePointer, torch::Tensor edgeList,
                                        torch::Tensor blockPartition, torch::Tensor edgeToColumn, torch::Tensor edgeToRow) {
  CHECK_INPUT(input); CHECK_INPUT(nodePointer); CHECK_INPUT(edgeList);
  CHECK_INPUT(blockPartition); CHECK_INPUT(edgeToColumn); CHECK_INPUT(edgeToRow);
  TORCH_CHECK(input.scalar_type() == torch::kFloat32, "expected scalar type Float");
  check_rows(input, nodePointer);
  DeviceGuard guard(input.device());
  auto output = torch::empty_like(input);                          // fully overwritten by the kernels (every row is a node's)
  if (input.numel() == 0) return {output};
  auto* plan = plan_for(nodePointer, edgeList, blockPartition, edgeToColumn, edgeToRow);
  const int D = (int)input.size(1);
  Workspace ws(plan, D, input);
  tcgnn_check(tcgnn_spmm(plan, input.data_ptr<float>(), output.data_ptr<float>(), D, ws.ptr, ws.bytes, current_stream(input)), "tcgnn_spmm");
  return {output};
}

// TCGNN.cpp:93-118 (edgeAttention sits between edgeList and blockPartition, call site gnn_conv.py:132)
std::vector<torch::Tensor> spmm_forward_AGNN(torch::Tensor input, torch::Tensor nodePointer, torch::Tensor edgeList,
                                             torch::Tensor edgeAttention, torch::Tensor blockPartition, torch::Tensor edgeToColumn,
                                             torch::Tensor edgeToRow) {
  CHECK_INPUT(input); CHECK_INPUT(nodePointer); CHECK_INPUT(edgeList); CHECK_INPUT(edgeAttention);
  CHECK_INPUT(blockPartition); CHECK_INPUT(edgeToColumn); CHECK_INPUT(edgeToRow);
  TORCH_CHECK(input.scalar_type() == torch::kFloat32 && edgeAttention.scalar_type() == torch::kFloat32, "expected scalar type Float");
  TORCH_CHECK(edgeAttention.numel() >= edgeList.numel(), "edgeAttention holds fewer values than there are edges");
  check_rows(input, nodePointer);
  DeviceGuard guard(input.device());
  auto output = torch::empty_like(input);
  if (input.numel() == 0) return {output};
  auto* plan = plan_for(nodePointer, edgeList, blockPartition, edgeToColumn, edgeToRow);
  const int D = (int)input.size(1);
  Workspace ws(plan, D, input);
  // every "head" launch of the reference reads row 0 of edgeAttention and overwrites the same output (TCGNN_kernel.cu:253-268, :529)
  tcgnn_check(tcgnn_spmm_val(plan, input.data_ptr<float>(), edgeAttention.data_ptr<float>(), output.data_ptr<float>(), D, ws.ptr, ws.bytes,
                             current_stream(input)), "tcgnn_spmm_val");
  return {output};
}

// TCGNN.cpp:126-150
std::vector<torch::Tensor> sddmm_forward(torch::Tensor input, torch::Tensor nodePointer, torch::Tensor edgeList,
                                         torch::Tensor blockPartition, torch::Tensor edgeToColumn, torch::Tensor edgeToRow) {
  CHECK_INPUT(input); CHECK_INPUT(nodePointer); CHECK_INPUT(edgeList);
  CHECK_INPUT(blockPartition); CHECK_INPUT(edgeToColumn); CHECK_INPUT(edgeToRow);
  TORCH_CHECK(input.scalar_type() == torch::kFloat32, "expected scalar type Float");
  check_rows(input, nodePointer);
  DeviceGuard guard(input.device());
  auto ef = torch::empty({edgeList.size(0)}, input.options());
  if (edgeList.size(0) == 0) return {ef};
  if (input.size(1) == 0) return {ef.zero_()};
  auto* plan = plan_for(nodePointer, edgeList, blockPartition, edgeToColumn, edgeToRow);
  const int D = (int)input.size(1);
  Workspace ws(plan, D, input);
  tcgnn_check(tcgnn_sddmm(plan, input.data_ptr<float>(), ef.data_ptr<float>(), D, ws.ptr, ws.bytes, current_stream(input)), "tcgnn_sddmm");
  return {ef};
}

// TCGNN.cpp:172-226: host tensors, outputs written in place, two lines on C stdout
void preprocess(torch::Tensor edgeList, torch::Tensor nodePointer, int num_nodes, int blockSize_h, int blockSize_w,
                torch::Tensor blockPartition, torch::Tensor edgeToColumn, torch::Tensor edgeToRow) {
  for (auto* t : {&edgeList, &nodePointer, &blockPartition, &edgeToColumn, &edgeToRow}) {
    TORCH_CHECK(!t->is_cuda(), "preprocess takes CPU tensors (preprocess_gpu takes device tensors)");
    TORCH_CHECK(t->is_contiguous(), "metadata tensors must be contiguous");
    TORCH_CHECK(t->scalar_type() == torch::kInt32, "expected scalar type Int");
  }
  TORCH_CHECK(nodePointer.numel() >= (int64_t)num_nodes + 1, "nodePointer must hold num_nodes + 1 entries");
  const int64_t E = nodePointer.data_ptr<int>()[num_nodes];
  TORCH_CHECK(edgeList.numel() >= E && edgeToColumn.numel() >= E && edgeToRow.numel() >= E, "edge arrays are shorter than nodePointer[num_nodes]");
  int64_t tc_blocks = 0;
  tcgnn_check(tcgnn_preprocess(edgeList.data_ptr<int>(), nodePointer.data_ptr<int>(), num_nodes, blockSize_h, blockSize_w,
                               blockPartition.data_ptr<int>(), blockPartition.numel(), edgeToColumn.data_ptr<int>(),
                               edgeToRow.data_ptr<int>(), &tc_blocks, 0), "tcgnn_preprocess");
  printf("TC_Blocks:\t%lld\nExp_Edges:\t%lld\n", (long long)tc_blocks, (long long)tc_blocks * 8 * 16);   // TCGNN.cpp:225
  fflush(stdout);
}

// TCGNN.cpp:229-256 (a stub in the reference: fill_window is empty, TCGNN_kernel.cu:42-80)
void preprocess_gpu(torch::Tensor edgeList, torch::Tensor nodePointer, int num_nodes, int blockSize_h, int blockSize_w,
                    torch::Tensor blockPartition, torch::Tensor edgeToColumn, torch::Tensor edgeToRow) {
  CHECK_INPUT(edgeList); CHECK_INPUT(nodePointer); CHECK_INPUT(blockPartition); CHECK_INPUT(edgeToColumn); CHECK_INPUT(edgeToRow);
  for (auto* t : {&edgeList, &nodePointer, &blockPartition, &edgeToColumn, &edgeToRow})
    TORCH_CHECK(t->scalar_type() == torch::kInt32, "expected scalar type Int");
  TORCH_CHECK(nodePointer.numel() >= (int64_t)num_nodes + 1, "nodePointer must hold num_nodes + 1 entries");
  TORCH_CHECK(edgeToColumn.numel() >= edgeList.numel() && edgeToRow.numel() >= edgeList.numel(), "edgeToColumn / edgeToRow are shorter than edgeList");
  DeviceGuard guard(edgeList.device());
  int64_t tc_blocks = 0;
  // scratch from torch's caching allocator: the library call allocates nothing and synchronises once (tcgnn_preprocess_gpu_ws)
  size_t need = 0;
  tcgnn_check(tcgnn_preprocess_gpu_workspace_bytes(num_nodes, edgeList.numel(), blockSize_h, &need), "tcgnn_preprocess_gpu_workspace_bytes");
  torch::Tensor ws = torch::empty({(int64_t)std::max<size_t>(need, 256)}, torch::TensorOptions().dtype(torch::kUInt8).device(edgeList.device()));
  tcgnn_check(tcgnn_preprocess_gpu_ws(edgeList.data_ptr<int>(), nodePointer.data_ptr<int>(), num_nodes, edgeList.numel(), blockSize_h, blockSize_w,
                                      blockPartition.data_ptr<int>(), blockPartition.numel(), edgeToColumn.data_ptr<int>(),
                                      edgeToRow.data_ptr<int>(), ws.data_ptr(), (size_t)ws.numel(), &tc_blocks, current_stream(edgeList)), "tcgnn_preprocess_gpu_ws");
  printf("TC_Blocks:\t%lld\nExp_Edges:\t%lld\n", (long long)tc_blocks, (long long)tc_blocks * 8 * 16);
  fflush(stdout);
}

void clear_plan_cache() {
  for (auto& e : plan_cache()) {
    if (!e.keep.empty() && e.keep[0].is_cuda()) { DeviceGuard guard(e.keep[0].device()); (void)hipDeviceSynchronize(); }   // nothing may still be reading the plan
    tcgnn_plan_destroy(e.plan);
  }
  plan_cache().clear();
}

// TCGNN.cpp:260-272
PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("preprocess", &preprocess, "Preprocess Step (CPU)");
  m.def("preprocess_gpu", &preprocess_gpu, "Preprocess Step (CUDA)");
  m.def("forward", &spmm_forward, "TC-GNN SPMM forward (CUDA)");
  m.def("forward_ef", &sddmm_forward, "TC-GNN SDDMM forward (CUDA)");
  m.def("forward_AGNN", &spmm_forward_AGNN, "TC-GNN SPMM (AGNN) forward (CUDA)");
  m.def("backward", &spmm_forward, "TC-GNN SPMM backward (CUDA)");
  m.def("backward_ef", &sddmm_forward, "TC-GNN SDDMM backward (CUDA)");
  m.def("clear_plan_cache", &clear_plan_cache, "release the device plans (not in the reference)");
}
