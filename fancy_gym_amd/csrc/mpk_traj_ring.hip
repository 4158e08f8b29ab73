// k_traj_ring / k_traj_burst behind one template launcher per MP type (-DMPK_MP_UNIT=0 promp, 1 dmp, 2 prodmp: one translation
// unit each, like mpk_traj_family.hip; without MPK_MP_UNIT -- the single-unit build mpk_kernels.hip -- all three are instantiated).
#include "mpk_traj_ring.h"

namespace mpk {

#ifndef MPK_DEVICE_ONLY
template <int MP, int CT>
static int launch_ring_t(const TrajArgs& ta, const ActArgs& aa, int blocks, size_t lds, void* stream) {
    if constexpr (MP == MPK_MP_DMP) {
        (void)ta; (void)aa; (void)blocks; (void)lds; (void)stream;
        set_error("internal: k_traj_ring is promp / prodmp");
        return MPK_EINVAL;
    } else if constexpr (CT >= 3) {
        // closed loop: producers + store engine + consumer waves (one group's recurrence per lane quarter); the DoF count is compiled
        // in (the launcher sends 5 or 7 DoF with <= 8 contraction columns here and everything else to k_traj_quad / duo / pipe)
        const dim3 g(blocks), br((unsigned)(ta.ring_np + ta.ring_ns + ta.ring_nc * (1 + ta.ring_aw)) * 64u);
        const int km = ta.c.KP / 4;
        auto go = [&](auto kern) {
            if (lds > 48 * 1024) (void)allow_full_lds(kern);
            hipLaunchKernelGGL(kern, g, br, lds, (hipStream_t)stream, ta, aa);
        };
        if (km > 2 || (ta.c.D != 5 && ta.c.D != 7)) { set_error("internal: closed-loop k_traj_ring takes 5 or 7 DoF"); return MPK_EINVAL; }
        if (ta.c.D == 7) { if (km == 1) go(k_traj_ring<MP, CT, 1, 7>); else go(k_traj_ring<MP, CT, 2, 7>); }
        else { if (km == 1) go(k_traj_ring<MP, CT, 1, 5>); else go(k_traj_ring<MP, CT, 2, 5>); }
        MPK_LAUNCH_CHECK();
        return MPK_OK;
    } else {
    const dim3 g(blocks);
    hipStream_t s = (hipStream_t)stream;
    const int km = ta.c.KP / 4;
    if (ta.burst == 1) {
        if constexpr (MP != MPK_MP_DMP && CT < 3) {
            const dim3 bb((unsigned)(ta.ring_m * ta.ring_np) * 64u);
            auto go = [&](auto kern) {
                if (lds > 48 * 1024) (void)allow_full_lds(kern);
                hipLaunchKernelGGL(kern, g, bb, lds, s, ta, aa);
            };
            switch (km) {
                case 1: go(k_traj_burst<MP, CT, 1>); break;
                case 2: go(k_traj_burst<MP, CT, 2>); break;
                case 3: go(k_traj_burst<MP, CT, 3>); break;
                default: go(k_traj_burst<MP, CT, 4>); break;
            }
        } else {
            set_error("internal: k_traj_burst is open loop, promp / prodmp");
            return MPK_EINVAL;
        }
        MPK_LAUNCH_CHECK();
        return MPK_OK;
    }
    if (ta.burst == 2) {
        // k_traj_flat with the DoF count compiled in (the launcher sends only D = 5 / 7 with <= 8 columns here)
        const dim3 bf(256);
        auto gof = [&](auto kern) {
            if (lds > 48 * 1024) (void)allow_full_lds(kern);
            hipLaunchKernelGGL(kern, g, bf, lds, s, ta, aa);
        };
        if (ta.c.D == 7) { if (km == 1) gof(k_traj_flat_d<MP, CT, 1, 7>); else gof(k_traj_flat_d<MP, CT, 2, 7>); }
        else { if (km == 1) gof(k_traj_flat_d<MP, CT, 1, 5>); else gof(k_traj_flat_d<MP, CT, 2, 5>); }
        MPK_LAUNCH_CHECK();
        return MPK_OK;
    }
    const dim3 br((unsigned)(ta.ring_np + ta.ring_ns) * 64u);
    auto go = [&](auto kern) {
        if (lds > 48 * 1024) (void)allow_full_lds(kern);
        hipLaunchKernelGGL(kern, g, br, lds, s, ta, aa);
    };
    // the DoF count as a compile-time constant for the shapes the reference registers MP environments with (5 and 7 DoF) and the
    // contraction lengths they have (<= 8 columns); everything else takes the generic loops
    auto by_km = [&](auto dc) {
        constexpr int DC = decltype(dc)::value;
        if (km == 1) go(k_traj_ring<MP, CT, 1, DC>);
        else go(k_traj_ring<MP, CT, 2, DC>);
    };
    if (km <= 2 && ta.c.D == 7) by_km(std::integral_constant<int, 7>());
    else if (km <= 2 && ta.c.D == 5) by_km(std::integral_constant<int, 5>());
    else {
        switch (km) {
            case 1: go(k_traj_ring<MP, CT, 1, 0>); break;
            case 2: go(k_traj_ring<MP, CT, 2, 0>); break;
            case 3: go(k_traj_ring<MP, CT, 3, 0>); break;
            default: go(k_traj_ring<MP, CT, 4, 0>); break;
        }
    }
    MPK_LAUNCH_CHECK();
    return MPK_OK;
    }
}

template <int MP>
int launch_traj_ring(const TrajArgs& ta, const ActArgs& aa, int ct, int blocks, size_t lds, void* stream) {
    if constexpr (MP != MPK_MP_DMP) {
        switch (ct) {
            case MPK_CTRL_MOTOR: return launch_ring_t<MP, MPK_CTRL_MOTOR>(ta, aa, blocks, lds, stream);
            case MPK_CTRL_VELOCITY: return launch_ring_t<MP, MPK_CTRL_VELOCITY>(ta, aa, blocks, lds, stream);
            case MPK_CTRL_POSITION: return launch_ring_t<MP, MPK_CTRL_POSITION>(ta, aa, blocks, lds, stream);
            case 3 + MPK_CTRL_MOTOR: return launch_ring_t<MP, 3 + MPK_CTRL_MOTOR>(ta, aa, blocks, lds, stream);
            case 3 + MPK_CTRL_VELOCITY: return launch_ring_t<MP, 3 + MPK_CTRL_VELOCITY>(ta, aa, blocks, lds, stream);
            case 3 + MPK_CTRL_POSITION: return launch_ring_t<MP, 3 + MPK_CTRL_POSITION>(ta, aa, blocks, lds, stream);
            default: break;
        }
    }
    return launch_ring_t<MP, -1>(ta, aa, blocks, lds, stream);
}

#ifdef MPK_MP_UNIT
template int launch_traj_ring<MPK_MP_UNIT>(const TrajArgs&, const ActArgs&, int, int, size_t, void*);
#else
template int launch_traj_ring<MPK_MP_PROMP>(const TrajArgs&, const ActArgs&, int, int, size_t, void*);
template int launch_traj_ring<MPK_MP_DMP>(const TrajArgs&, const ActArgs&, int, int, size_t, void*);
template int launch_traj_ring<MPK_MP_PRODMP>(const TrajArgs&, const ActArgs&, int, int, size_t, void*);
#endif
#endif

}  // namespace mpk
