// Temporary: entry points not implemented yet return MI355X_EUNSUPPORTED.
#include "common.cuh"
using namespace mi355x;
#define STUB(name, ...) extern "C" int name(__VA_ARGS__) { set_error(#name ": not implemented yet"); return MI355X_EUNSUPPORTED; }
STUB(mi355x_paged_prefill_attention, void*, const void*, const void*, const void*, int, int, int, int, int, float, const int*, const int*, const int*, int, int, int64_t, int64_t, int64_t, int64_t, int, mi355x_stream)
