// Single-translation-unit build of every gfx950 kernel of libmpk.so: tools/dev/one_kernel.sh (one instantiation in seconds,
// -DMPK_DEVICE_ONLY) and the -DMPK_TRACE development builds (the trace buffer is one device variable).  The library itself is
// built from the files below as separate translation units, in parallel (__graft_entry__.py):
//   mpk_dev.h            scalar device helpers shared by everything
//   mpk_tile.h           the 16 x 16 tile machinery (arguments, lane maps, step chains, epilogue, stores)
//   mpk_traj_{tiles,stream,flat,quad,pipe}.h   one shared-phase trajectory kernel family each
//   mpk_traj_family.hip  the families' template launcher, one unit per MP type (-DMPK_MP_UNIT=0..2)
//   mpk_traj_ring.h / .hip   k_traj_ring / k_traj_burst + their launcher, one unit per MP type
//   mpk_episode.hip      k_episode_return (verbose < 2 step: nothing per step stored), one unit per MP type
//   mpk_reward.h         SimpleReacher reward on float64 LDS images (rollout + episode kernels)
//   mpk_traj_launch.hip  k_build_shared + launch_traj_shared (kernel selection rule)
//   mpk_traj_wide.hip    k_traj_wide
//   mpk_traj_phase.hip   per-episode phase kernels
//   mpk_phase_fused.hip  per-episode phase: the fused entry points (actions, closed loop, replanning step, verbose < 2 step, validity gate)
//   mpk_rollout.hip      rollout kernels
//   mpk_misc.hip         integer state, reset, gather, validity, self-tests, trace readout
#define MPK_AMALGAMATED 1
#include "mpk_traj_family.hip"
#include "mpk_traj_ring.hip"
#include "mpk_episode.hip"
#include "mpk_traj_launch.hip"
#include "mpk_traj_wide.hip"
#include "mpk_traj_phase.hip"
#include "mpk_phase_fused.hip"
#include "mpk_rollout.hip"
#include "mpk_misc.hip"
