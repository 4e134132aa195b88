// graph.hip -- hipGraph capture / replay of static launch sequences (include/dwg_graph.h).
#include "dwg_common.h"
#include "../../include/dwg_graph.h"

namespace {
struct Graph { hipGraph_t graph; hipGraphExec_t exec; };
}

extern "C" {

int dwg_graph_begin_capture(dwg_stream_t stream) {
    if (!stream) return DWG_E_ARG;   // the legacy default stream cannot be captured
    if (hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal) != hipSuccess) return DWG_E_LAUNCH;
    return DWG_OK;
}

int dwg_graph_end_capture(dwg_stream_t stream, dwg_graph_t* out) {
    if (!stream || !out) return DWG_E_ARG;
    Graph* g = new Graph{nullptr, nullptr};
    if (hipStreamEndCapture((hipStream_t)stream, &g->graph) != hipSuccess || !g->graph) { delete g; return DWG_E_LAUNCH; }
    if (hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0) != hipSuccess) { hipGraphDestroy(g->graph); delete g; return DWG_E_LAUNCH; }
    *out = g;
    return DWG_OK;
}

int dwg_graph_launch(dwg_graph_t graph, dwg_stream_t stream) {
    if (!graph) return DWG_E_ARG;
    if (hipGraphLaunch(((Graph*)graph)->exec, (hipStream_t)stream) != hipSuccess) return DWG_E_LAUNCH;
    return DWG_OK;
}

int dwg_graph_destroy(dwg_graph_t graph) {
    if (!graph) return DWG_OK;
    Graph* g = (Graph*)graph;
    hipGraphExecDestroy(g->exec); hipGraphDestroy(g->graph);
    delete g;
    return DWG_OK;
}

}  // extern "C"
