// The shared-phase trajectory kernel families (k_traj_tiles / split / stream / flat / quad / pipe) behind one template
// launcher per MP type.  Built once per MP type (-DMPK_MP_UNIT=0 promp, 1 dmp, 2 prodmp) so that the ~300 instantiations
// compile on several cores; without MPK_MP_UNIT (the single-unit build mpk_kernels.hip) all three are instantiated here.
#include "mpk_traj_tiles.h"
#include "mpk_traj_stream.h"
#include "mpk_traj_flat.h"
#include "mpk_traj_quad.h"
#include "mpk_traj_pipe.h"

namespace mpk {

#ifndef MPK_DEVICE_ONLY
template <int MP, int CT>
static int launch_traj_t(const TrajArgs& ta, const ActArgs& aa, bool stream_mode, bool write_through, bool bulk,
                         int quad, int blocks, size_t lds, void* stream, bool split = false, bool pipe = false) {
    const dim3 g(blocks), b(256);
    if (pipe) {
        if constexpr (MP != MPK_MP_DMP && CT >= 3) {
            const dim3 b5(320);
            hipStream_t s5 = (hipStream_t)stream;
            auto gop = [&](auto lean) {
                constexpr bool LEAN = decltype(lean)::value;
                switch (ta.c.KP / 4) {
                    case 1: hipLaunchKernelGGL((k_traj_pipe<MP, CT, 1, LEAN>), g, b5, lds, s5, ta, aa); break;
                    case 2: hipLaunchKernelGGL((k_traj_pipe<MP, CT, 2, LEAN>), g, b5, lds, s5, ta, aa); break;
                    case 3: hipLaunchKernelGGL((k_traj_pipe<MP, CT, 3, LEAN>), g, b5, lds, s5, ta, aa); break;
                    default: hipLaunchKernelGGL((k_traj_pipe<MP, CT, 4, LEAN>), g, b5, lds, s5, ta, aa); break;
                }
            };
            auto gog = [&]() {                          // the validity gate: the consumer's chain with the GATE hook (no LEAN form)
                switch (ta.c.KP / 4) {
                    case 1: hipLaunchKernelGGL((k_traj_pipe<MP, CT, 1, false, true>), g, b5, lds, s5, ta, aa); break;
                    case 2: hipLaunchKernelGGL((k_traj_pipe<MP, CT, 2, false, true>), g, b5, lds, s5, ta, aa); break;
                    case 3: hipLaunchKernelGGL((k_traj_pipe<MP, CT, 3, false, true>), g, b5, lds, s5, ta, aa); break;
                    default: hipLaunchKernelGGL((k_traj_pipe<MP, CT, 4, false, true>), g, b5, lds, s5, ta, aa); break;
                }
            };
            if (ta.gate_valid) gog();
            else if (ta.lean) gop(std::true_type()); else gop(std::false_type());
        }
        MPK_LAUNCH_CHECK();
        return MPK_OK;
    }
    // tile-major / split: no dynamic LDS of their own; `lds` then is the occupancy-experiment padding ("lds_pad" option)
    const size_t pad = (!stream_mode || split) ? lds : 0;
    hipStream_t s = (hipStream_t)stream;
    const int km = ta.c.KP / 4;
    if (ta.flat_img > 0) {
        if constexpr (MP != MPK_MP_DMP && CT < 3) {
            auto go = [&](auto kern) {
                if (lds > 48 * 1024) (void)allow_full_lds(kern);
                hipLaunchKernelGGL(kern, g, b, lds, s, ta, aa);
            };
            switch (km) {
                case 1: go(k_traj_flat<MP, CT, 1>); break;
                case 2: go(k_traj_flat<MP, CT, 2>); break;
                case 3: go(k_traj_flat<MP, CT, 3>); break;
                default: go(k_traj_flat<MP, CT, 4>); break;
            }
        }
        MPK_LAUNCH_CHECK();
        return MPK_OK;
    }
    if (split) {
        if constexpr (MP != MPK_MP_DMP && CT >= 3) {
            if (write_through) {
                switch (km) {
                    case 1: hipLaunchKernelGGL((k_traj_split<MP, CT, 1, true>), g, b, pad, s, ta, aa); break;
                    case 2: hipLaunchKernelGGL((k_traj_split<MP, CT, 2, true>), g, b, pad, s, ta, aa); break;
                    case 3: hipLaunchKernelGGL((k_traj_split<MP, CT, 3, true>), g, b, pad, s, ta, aa); break;
                    default: hipLaunchKernelGGL((k_traj_split<MP, CT, 4, true>), g, b, pad, s, ta, aa); break;
                }
            } else {
                switch (km) {
                    case 1: hipLaunchKernelGGL((k_traj_split<MP, CT, 1, false>), g, b, pad, s, ta, aa); break;
                    case 2: hipLaunchKernelGGL((k_traj_split<MP, CT, 2, false>), g, b, pad, s, ta, aa); break;
                    case 3: hipLaunchKernelGGL((k_traj_split<MP, CT, 3, false>), g, b, pad, s, ta, aa); break;
                    default: hipLaunchKernelGGL((k_traj_split<MP, CT, 4, false>), g, b, pad, s, ta, aa); break;
                }
            }
        }
    } else if (stream_mode && quad) {
        if constexpr (MP == MPK_MP_DMP || CT >= 3) {
            if (quad == 1) {
                switch (km) {
                    case 1: hipLaunchKernelGGL((k_traj_quad<MP, CT, 1, 1>), g, b, lds, s, ta, aa); break;
                    case 2: hipLaunchKernelGGL((k_traj_quad<MP, CT, 2, 1>), g, b, lds, s, ta, aa); break;
                    case 3: hipLaunchKernelGGL((k_traj_quad<MP, CT, 3, 1>), g, b, lds, s, ta, aa); break;
                    default: hipLaunchKernelGGL((k_traj_quad<MP, CT, 4, 1>), g, b, lds, s, ta, aa); break;
                }
            } else if (quad == 2) {
                switch (km) {
                    case 1: hipLaunchKernelGGL((k_traj_quad<MP, CT, 1, 2>), g, b, lds, s, ta, aa); break;
                    case 2: hipLaunchKernelGGL((k_traj_quad<MP, CT, 2, 2>), g, b, lds, s, ta, aa); break;
                    case 3: hipLaunchKernelGGL((k_traj_quad<MP, CT, 3, 2>), g, b, lds, s, ta, aa); break;
                    default: hipLaunchKernelGGL((k_traj_quad<MP, CT, 4, 2>), g, b, lds, s, ta, aa); break;
                }
            } else {
                switch (km) {
                    case 1: hipLaunchKernelGGL((k_traj_quad<MP, CT, 1, 4>), g, b, lds, s, ta, aa); break;
                    case 2: hipLaunchKernelGGL((k_traj_quad<MP, CT, 2, 4>), g, b, lds, s, ta, aa); break;
                    case 3: hipLaunchKernelGGL((k_traj_quad<MP, CT, 3, 4>), g, b, lds, s, ta, aa); break;
                    default: hipLaunchKernelGGL((k_traj_quad<MP, CT, 4, 4>), g, b, lds, s, ta, aa); break;
                }
            }
        }
    } else if (stream_mode) {
        if (bulk) {
            // more than 48 KB of dynamic LDS only happens with the "lds_pad" occupancy knob (one workgroup per CU)
            auto big = [&](auto kern) {
                if (lds > 48 * 1024) (void)allow_full_lds(kern);
            };
            switch (km) {
                case 1: big(k_traj_stream<MP, CT, 1, true>); hipLaunchKernelGGL((k_traj_stream<MP, CT, 1, true>), g, b, lds, s, ta, aa); break;
                case 2: big(k_traj_stream<MP, CT, 2, true>); hipLaunchKernelGGL((k_traj_stream<MP, CT, 2, true>), g, b, lds, s, ta, aa); break;
                case 3: big(k_traj_stream<MP, CT, 3, true>); hipLaunchKernelGGL((k_traj_stream<MP, CT, 3, true>), g, b, lds, s, ta, aa); break;
                default: big(k_traj_stream<MP, CT, 4, true>); hipLaunchKernelGGL((k_traj_stream<MP, CT, 4, true>), g, b, lds, s, ta, aa); break;
            }
        } else {
            switch (km) {
                case 1: hipLaunchKernelGGL((k_traj_stream<MP, CT, 1, false>), g, b, lds, s, ta, aa); break;
                case 2: hipLaunchKernelGGL((k_traj_stream<MP, CT, 2, false>), g, b, lds, s, ta, aa); break;
                case 3: hipLaunchKernelGGL((k_traj_stream<MP, CT, 3, false>), g, b, lds, s, ta, aa); break;
                default: hipLaunchKernelGGL((k_traj_stream<MP, CT, 4, false>), g, b, lds, s, ta, aa); break;
            }
        }
    } else {
        if constexpr (MP != MPK_MP_DMP && CT < 3) {
            const dim3 bt((unsigned)ta.wpb * 64u);
            if (write_through) {
                switch (km) {
                    case 1: hipLaunchKernelGGL((k_traj_tiles<MP, CT, 1, true>), g, bt, pad, s, ta, aa); break;
                    case 2: hipLaunchKernelGGL((k_traj_tiles<MP, CT, 2, true>), g, bt, pad, s, ta, aa); break;
                    case 3: hipLaunchKernelGGL((k_traj_tiles<MP, CT, 3, true>), g, bt, pad, s, ta, aa); break;
                    default: hipLaunchKernelGGL((k_traj_tiles<MP, CT, 4, true>), g, bt, pad, s, ta, aa); break;
                }
            } else {
                switch (km) {
                    case 1: hipLaunchKernelGGL((k_traj_tiles<MP, CT, 1, false>), g, bt, pad, s, ta, aa); break;
                    case 2: hipLaunchKernelGGL((k_traj_tiles<MP, CT, 2, false>), g, bt, pad, s, ta, aa); break;
                    case 3: hipLaunchKernelGGL((k_traj_tiles<MP, CT, 3, false>), g, bt, pad, s, ta, aa); break;
                    default: hipLaunchKernelGGL((k_traj_tiles<MP, CT, 4, false>), g, bt, pad, s, ta, aa); break;
                }
            }
        }
    }
    MPK_LAUNCH_CHECK();
    return MPK_OK;
}
#endif  // MPK_DEVICE_ONLY

#ifndef MPK_DEVICE_ONLY
template <int MP>
int launch_traj_ct(const TrajArgs& ta, const ActArgs& aa, int ct, bool stream_mode, bool write_through,
                   bool bulk, int quad, int blocks, size_t lds, void* stream, bool split, bool pipe) {
    if constexpr (MP != MPK_MP_DMP) {
        if (pipe) {
            switch (ct) {
                case 3 + MPK_CTRL_MOTOR: return launch_traj_t<MP, 3 + MPK_CTRL_MOTOR>(ta, aa, true, false, false, 0, blocks, lds, stream, false, true);
                case 3 + MPK_CTRL_VELOCITY: return launch_traj_t<MP, 3 + MPK_CTRL_VELOCITY>(ta, aa, true, false, false, 0, blocks, lds, stream, false, true);
                default: return launch_traj_t<MP, 3 + MPK_CTRL_POSITION>(ta, aa, true, false, false, 0, blocks, lds, stream, false, true);
            }
        }
        if (split) {
            switch (ct) {
                case 3 + MPK_CTRL_MOTOR: return launch_traj_t<MP, 3 + MPK_CTRL_MOTOR>(ta, aa, false, write_through, false, 0, blocks, lds, stream, true);
                case 3 + MPK_CTRL_VELOCITY: return launch_traj_t<MP, 3 + MPK_CTRL_VELOCITY>(ta, aa, false, write_through, false, 0, blocks, lds, stream, true);
                default: return launch_traj_t<MP, 3 + MPK_CTRL_POSITION>(ta, aa, false, write_through, false, 0, blocks, lds, stream, true);
            }
        }
        switch (ct) {
            case MPK_CTRL_MOTOR: return launch_traj_t<MP, MPK_CTRL_MOTOR>(ta, aa, stream_mode, write_through, bulk, quad, blocks, lds, stream);
            case MPK_CTRL_VELOCITY: return launch_traj_t<MP, MPK_CTRL_VELOCITY>(ta, aa, stream_mode, write_through, bulk, quad, blocks, lds, stream);
            case MPK_CTRL_POSITION: return launch_traj_t<MP, MPK_CTRL_POSITION>(ta, aa, stream_mode, write_through, bulk, quad, blocks, lds, stream);
            case 3 + MPK_CTRL_MOTOR: return launch_traj_t<MP, 3 + MPK_CTRL_MOTOR>(ta, aa, true, false, bulk, quad, blocks, lds, stream);
            case 3 + MPK_CTRL_VELOCITY: return launch_traj_t<MP, 3 + MPK_CTRL_VELOCITY>(ta, aa, true, false, bulk, quad, blocks, lds, stream);
            case 3 + MPK_CTRL_POSITION: return launch_traj_t<MP, 3 + MPK_CTRL_POSITION>(ta, aa, true, false, bulk, quad, blocks, lds, stream);
            default: break;
        }
    }
    return launch_traj_t<MP, -1>(ta, aa, stream_mode, write_through, bulk, quad, blocks, lds, stream);
}
#endif  // MPK_DEVICE_ONLY

#ifndef MPK_DEVICE_ONLY
#ifdef MPK_MP_UNIT
template int launch_traj_ct<MPK_MP_UNIT>(const TrajArgs&, const ActArgs&, int, bool, bool, bool, int, int, size_t, void*, bool, bool);
#else
template int launch_traj_ct<MPK_MP_PROMP>(const TrajArgs&, const ActArgs&, int, bool, bool, bool, int, int, size_t, void*, bool, bool);
template int launch_traj_ct<MPK_MP_DMP>(const TrajArgs&, const ActArgs&, int, bool, bool, bool, int, int, size_t, void*, bool, bool);
template int launch_traj_ct<MPK_MP_PRODMP>(const TrajArgs&, const ActArgs&, int, bool, bool, bool, int, int, size_t, void*, bool, bool);
#endif
#endif

}  // namespace mpk
