// mgm_pass2_dispatch.hip -- picks the per-LPL object of the second K3 build
// (mgm_pass2.hip is compiled once per LPL with -DMGM_P2_LPL=n).
#include "mgm_device.h"

#ifndef MGM_P2_C8_NL
#define MGM_P2_C8_NL 1
#endif
#ifndef MGM_P2_C8_NC
#define MGM_P2_C8_NC (16 - MGM_P2_C8_NL)
#endif
#ifndef MGM_P2_DEV
#define MGM_P2_DEV 0
#endif

namespace mgm {

// whether the second build carries its in-kernel timers / experiment switches (MGM_HIP_DEBUG_STATS, MGM_HIP_XFLAGS)
bool pass2_devtools() { return MGM_P2_DEV != 0; }
#ifndef MGM_P2_TIMELINE
#define MGM_P2_TIMELINE 0
#endif
bool pass2_timeline() { return MGM_P2_TIMELINE != 0; }

// lines per band of the second build (0 = this L is not supported by it)
int pass2_lines(int L, bool c8)
{
    if (L % 64) return 0;
    const int lpl = L / 64;
    if (c8 && c8_supported(L)) return lpl <= 4 ? MGM_P2_C8_NC : 7;  // compact costs need fewer loader waves
    if (lpl == 1 || lpl == 2 || lpl == 3 || lpl == 4) return 14;
    if (lpl == 6 || lpl == 8 || lpl == 12 || lpl == 16) return 7;  // (768 / 1024 labels: round 4)
    return 0;
}

hipError_t launch_pass2(const PassParams &p, int ntasks, bool fh, int wmode, hipStream_t s)
{
    switch (p.subv > 1 ? 4 : (p.L % 64 ? 0 : p.L / 64)) {
        case 1: return launch_pass2_lpl<1>(p, ntasks, fh, wmode, s);
        case 2: return launch_pass2_lpl<2>(p, ntasks, fh, wmode, s);
        case 3: return launch_pass2_lpl<3>(p, ntasks, fh, wmode, s);
        case 4: return launch_pass2_lpl<4>(p, ntasks, fh, wmode, s);
        case 6: return launch_pass2_lpl<6>(p, ntasks, fh, wmode, s);
        case 8: return launch_pass2_lpl<8>(p, ntasks, fh, wmode, s);
        case 12: return launch_pass2_lpl<12>(p, ntasks, fh, wmode, s);
        case 16: return launch_pass2_lpl<16>(p, ntasks, fh, wmode, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace mgm
