// Library-level entry points: version, error string, device check.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include "../../include/vg_kernels.h"

static thread_local char g_err[512] = "";

extern "C" void vg_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* vg_last_error(void) { return g_err; }

extern "C" int vg_version(void) { return 100; }  // 0.1.0

extern "C" int vg_init(int device) {
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) {
    vg_set_error("vg_init: hipGetDeviceProperties(%d) failed: %s", device, hipGetErrorString(e));
    return VG_ERR_ARG;
  }
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    vg_set_error("vg_init: device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    return VG_ERR_UNSUPPORTED;
  }
  return prop.multiProcessorCount;
}

extern "C" int vg_graph_node_counts(void* graph, int64_t* counts) {
  if (!graph || !counts) {
    vg_set_error("vg_graph_node_counts: null argument");
    return VG_ERR_ARG;
  }
  size_t n = 0;
  hipError_t e = hipGraphGetNodes((hipGraph_t)graph, nullptr, &n);
  if (e != hipSuccess) {
    vg_set_error("vg_graph_node_counts: hipGraphGetNodes failed: %s", hipGetErrorString(e));
    return VG_ERR_ARG;
  }
  counts[0] = counts[1] = counts[2] = 0;
  if (n == 0) return VG_OK;
  hipGraphNode_t* nodes = new hipGraphNode_t[n];
  e = hipGraphGetNodes((hipGraph_t)graph, nodes, &n);
  for (size_t i = 0; e == hipSuccess && i < n; ++i) {
    hipGraphNodeType t;
    e = hipGraphNodeGetType(nodes[i], &t);
    if (e != hipSuccess) break;
    counts[t == hipGraphNodeTypeKernel ? 0 : (t == hipGraphNodeTypeMemcpy ? 1 : 2)] += 1;
  }
  delete[] nodes;
  if (e != hipSuccess) {
    vg_set_error("vg_graph_node_counts: %s", hipGetErrorString(e));
    return VG_ERR_ARG;
  }
  return VG_OK;
}
