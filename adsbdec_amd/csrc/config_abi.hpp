// config_abi.hpp -- how adsb_create and adsb_multi_create read the caller's adsb_config (one rule, two callers).
//
// The struct is the CALLER's: cfg->struct_size says how much of it exists, cfg->abi which header wrote it.  ABI 5 moved
// members (the debug_* knobs left for adsb_debug_config, `abi` came in second place), so a struct of another ABI is
// refused by name -- a binary built against ABI <= 4 has its df18 where `abi` is and would otherwise be misread silently.
// Host-only code, no HIP.
#pragma once

#include <cstddef>
#include <cstring>

#include "../../include/adsbdec_amd_diag.h"

namespace adsb {

// cfg / dbg receive the library's defaults overlaid with what the caller's structs hold; cfg.debug is cleared (the knobs
// are COPIED: the caller's adsb_debug_config need not outlive the call).  Returns nullptr, or why the struct is refused.
inline const char *accept_config(const adsb_config *in, adsb_config &cfg, adsb_debug_config &dbg)
{
    adsb_config_init(&cfg, sizeof cfg);
    std::memset(&dbg, 0, sizeof dbg);
    dbg.struct_size = sizeof dbg;
    if (!in)
        return nullptr;
    if (in->struct_size < offsetof(adsb_config, device) + sizeof(int32_t) || in->struct_size > sizeof cfg)
        return "adsb_config.struct_size is not one this library knows";
    if (in->abi != ADSB_ABI_VERSION)
        return "adsb_config.abi is not this library's ADSB_ABI_VERSION (5): the caller was built against another layout of "
               "adsb_config (ABI <= 4 had its debug_* knobs inside the struct) and must be rebuilt against include/adsbdec_amd.h; "
               "adsb_config_default() / adsb_config_init() set the member";
    std::memcpy(&cfg, in, in->struct_size);
    cfg.struct_size = sizeof cfg;
    if (in->struct_size >= offsetof(adsb_config, debug) + sizeof(void *) && in->debug) {
        const adsb_debug_config *dc = static_cast<const adsb_debug_config *>(in->debug);
        if (dc->struct_size < 2 * sizeof(uint32_t) || dc->struct_size > sizeof dbg)
            return "adsb_debug_config.struct_size is not one this library knows";
        std::memcpy(&dbg, dc, dc->struct_size);
        dbg.struct_size = sizeof dbg;
    }
    cfg.debug = nullptr;
    return nullptr;
}

} // namespace adsb
