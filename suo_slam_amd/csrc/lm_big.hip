// The LM / bundle-adjustment kernel of csrc/lm.hip compiled for a 1024-thread workgroup (16 waves on one CU).
//
// A frame-sized graph (8 objects x ~15 keypoints) is latency-bound and runs best as 4 waves that share a CU with the
// CNN; the global SLAM graph (every 10 views: cameras x objects x keypoints = thousands of edges, lib/object_slam.py:
// 444-447,703-903) is throughput-bound in its per-edge loops -- residuals, Jacobians, block accumulation -- which
// stride over the workgroup.  Same source, same algorithm, four times the lanes.
#ifndef SUO_LM_THREADS
#define SUO_LM_THREADS 1024
#endif
#define SUO_LM_BIG 1
#define lm_kernel lm_kernel_big
#include "lm.hip"
