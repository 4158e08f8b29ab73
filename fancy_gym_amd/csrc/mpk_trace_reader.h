// -DMPK_TRACE development builds only (tools/dev/trace_*.py): the host-side reader of the stamp buffer.  Included ONCE per library:
// by mpk_misc.hip in a whole-library trace build (MPK_BUILD_AMALGAMATED), by the traced unit itself in a one-unit trace build
// (MPK_TRACE_UNIT=<file>.hip: that unit alone is compiled with -DMPK_TRACE, every other unit as shipped -- minutes instead of
// the amalgamated build's six).
#pragma once
// development builds only: fetch and clear the stamps (pairs of tag, shader clock)
extern "C" int mpk_debug_trace(long long* out, int cap) {
    // out: (tag, clock) pairs of the slots stamped since the last call, sorted by clock; returns their number
    long long raw[256];
    if (hipMemcpyFromSymbol(raw, HIP_SYMBOL(mpk::g_trace), sizeof(raw)) != hipSuccess) return -1;
    int n = 0;
    for (int t = 0; t < 256 && n < cap; ++t)
        if (raw[t] != 0) { out[2 * n] = t; out[2 * n + 1] = raw[t]; ++n; }
    for (int i = 1; i < n; ++i)
        for (int j = i; j > 0 && out[2 * j + 1] < out[2 * j - 1]; --j) {
            const long long t0 = out[2 * j], c0 = out[2 * j + 1];
            out[2 * j] = out[2 * j - 2]; out[2 * j + 1] = out[2 * j - 1];
            out[2 * j - 2] = t0; out[2 * j - 1] = c0;
        }
    static const long long zeros[256] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(mpk::g_trace), zeros, sizeof(zeros));
    return n;
}
