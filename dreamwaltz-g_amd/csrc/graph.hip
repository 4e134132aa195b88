// graph.hip -- hipGraph capture / replay of static launch sequences (include/dwg_graph.h).
#include "dwg_common.h"
#include "../../include/dwg_graph.h"

#include <mutex>
#include <vector>

namespace {
struct Graph { hipGraph_t graph; hipGraphExec_t exec; };
// events used for cross-stream ordering: a small ring, created on demand, never destroyed while the library is loaded
// (an event recorded into a capture must stay valid until the capture ends; re-recording an event in eager mode is fine,
//  a pending hipStreamWaitEvent keeps the state it saw)
std::mutex g_ev_mu;
std::vector<hipEvent_t> g_events;
size_t g_ev_next = 0;
hipEvent_t next_event() {
    std::lock_guard<std::mutex> lk(g_ev_mu);
    if (g_events.size() < 256) {
        hipEvent_t e;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
        g_events.push_back(e);
        return e;
    }
    hipEvent_t e = g_events[g_ev_next];
    g_ev_next = (g_ev_next + 1) % g_events.size();
    return e;
}
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

int dwg_stream_create(dwg_stream_t* out) {
    if (!out) return DWG_E_ARG;
    hipStream_t s;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return DWG_E_LAUNCH;
    *out = (dwg_stream_t)s;
    return DWG_OK;
}

int dwg_stream_destroy(dwg_stream_t stream) {
    if (!stream) return DWG_OK;
    return hipStreamDestroy((hipStream_t)stream) == hipSuccess ? DWG_OK : DWG_E_LAUNCH;
}

int dwg_stream_fork(dwg_stream_t from, dwg_stream_t to) {
    if (from == to) return DWG_OK;
    hipEvent_t e = next_event();
    if (!e) return DWG_E_LAUNCH;
    if (hipEventRecord(e, (hipStream_t)from) != hipSuccess) return DWG_E_LAUNCH;
    if (hipStreamWaitEvent((hipStream_t)to, e, 0) != hipSuccess) return DWG_E_LAUNCH;
    return DWG_OK;
}

}  // extern "C"
