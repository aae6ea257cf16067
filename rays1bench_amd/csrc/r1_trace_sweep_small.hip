// r1_trace_sweep_small.hip — the trace kernel's instantiations for one family (r1_trace_tu.inc says which); kernel and device functions: r1_trace.hpp
#define R1_TU_NAME sweep_small
#define R1_TU_BIG false
#define R1_TU_TREE 0
#include "r1_trace_tu.inc"
