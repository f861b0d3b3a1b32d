// gfx950 wavefront path tracer: generate / extend / shade / shadow / probe /
// finish kernels over compacted ray queues.
//
// One path per lane, one 64-lane wavefront per workgroup.  A path's state
// lives in SoA float4 arrays in HBM and travels between stages by slot index;
// rays travel in compacted queues written with one atomic per wavefront.  The
// closest-hit and any-hit walks keep a per-lane stack in LDS, lane-strided so
// that a push or pop is one conflict-free ds access.
//
// What each stage computes follows the reference integrators
// (CUDA-PathTracer/tracer/pathtracer.cu:822-1228) exactly, sample by sample:
// the same per-pixel minstd_rand stream, the same draw order, the same
// arithmetic in the same order (built with -ffp-contract=off against
// fpspec.h), so an image is a pure function of (scene, pixel, iteration).
#include <hip/hip_runtime.h>

#include "../fpspec.h"
#include "device_scene.h"

// The kernels, part by part (each file opens namespace caster itself):
#include "kernels/common.h"
#include "kernels/traverse.h"
#include "kernels/trace.h"
#include "kernels/shading.h"
#include "kernels/bsdf.h"
#include "kernels/path.h"
#include "kernels/volpath.h"
#include "kernels/lighttrace.h"
#include "kernels/bdpt.h"
#include "kernels/volpath_queues.h"
#include "kernels/volpath_hetero.h"
#include "kernels/film.h"
#include "kernels/hooks.h"
