/*
 * include/dwg_graph.h -- hipGraph capture / replay of a static sequence of dwg_* launches.
 * The denoiser and VAE plans are launch-bound (hundreds of small kernels per step): they are recorded once with
 * hipStreamBeginCapture on a caller-provided (non-default) stream and replayed with one hipGraphLaunch per step.
 * No reference counterpart (the reference launches eagerly through PyTorch).
 */
#ifndef DWG_GRAPH_H
#define DWG_GRAPH_H
#include "dwg_types.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef void* dwg_graph_t;
/* Begin capturing everything subsequently enqueued on `stream` (must not be the legacy default stream). */
int dwg_graph_begin_capture(dwg_stream_t stream);
/* End the capture and instantiate an executable graph. */
int dwg_graph_end_capture(dwg_stream_t stream, dwg_graph_t* graph_out);
int dwg_graph_launch(dwg_graph_t graph, dwg_stream_t stream);
int dwg_graph_destroy(dwg_graph_t graph);

/* Side streams for plans with independent branches (e.g. the ControlNet encoder next to the UNet encoder: neither fills 256
 * CUs at the 16x16 / 8x8 latent levels, together they do).  dwg_stream_create makes a non-blocking stream owned by the
 * library.  dwg_stream_fork(from, to) orders `to` after everything enqueued on `from` so far (event record + stream wait);
 * called with a capturing `from` it pulls `to` into the capture, and fork(side, main) joins it back -- the captured graph
 * then holds the branches as parallel paths. */
int dwg_stream_create(dwg_stream_t* stream_out);
int dwg_stream_destroy(dwg_stream_t stream);
int dwg_stream_fork(dwg_stream_t from, dwg_stream_t to);
#ifdef __cplusplus
}
#endif
#endif
