// Dispatch record (hgs_dispatch_read): which kernel template instance every launch site of the engine actually
// launched.  engine.hip chooses among some twenty variants by shape, precision, slot count, list density and options; the
// results of neighbouring variants often agree to the last bit, so a test that only compares numbers cannot tell when
// the dispatcher routes a case to another kernel.  Every launcher therefore notes its instance here, at the one place
// where the template arguments are spelled out, and the tests assert the record next to the results.
//
// Cost per launch: one thread-local load and a pointer comparison over a handful of entries.
#pragma once
#include <string>
#include <vector>

namespace hgs {

// run-time properties of a launch that no template argument shows
enum : unsigned {
    DF_LIST = 1u,        // walks a column / tile list (sparse targets)
    DF_LOAD_MASK = 2u,   // row kernel: loads only the active columns
    DF_STORE_MASK = 4u,  // row kernel: stores only the active (or dilated) columns
    DF_XMAP = 8u,        // passes of a tile grouped per XCD (col_xmap / list_xmap)
    DF_BATCH = 16u,      // grid.y > 1
    DF_STATS = 32u,      // in-pass statistics requested (ColArgs::do_stats)
    DF_NF_OUT = 64u,     // complex nearfield kept (MultiplaneHologram)
};

struct DispatchSite {
    const char* family;   // kernel name
    const char* params;   // names of its template parameters, comma separated
    const char* pretty;   // __PRETTY_FUNCTION__ of dispatch_site<...>: carries the argument values
};

struct DispatchLog {
    struct Ent { const DispatchSite* site; unsigned flags; unsigned long long count; };
    std::vector<Ent> v;
    void hit(const DispatchSite* s, unsigned flags) {
        for (auto& e : v)
            if (e.site == s && e.flags == flags) { ++e.count; return; }
        v.push_back({s, flags, 1ull});
    }
    // "family<P0=v0,P1=v1,...> flag flag\tcount\n" per entry
    std::string text() const;
};

// the log of the engine whose C-ABI call is running on this thread (engine.hip sets it in every entry point)
extern thread_local DispatchLog* g_dispatch;

template <typename Fam, typename R, auto... V> inline const DispatchSite* dispatch_site() {
    static const DispatchSite s{Fam::name, Fam::params, __PRETTY_FUNCTION__};
    return &s;
}
inline void dispatch_note(const DispatchSite* s, unsigned flags = 0) {
    if (g_dispatch) g_dispatch->hit(s, flags);
}

// kernel families (template parameter names in declaration order, the element type first)
#define HGS_FAMILY(tag, kname, plist) struct tag { static constexpr const char* name = kname; static constexpr const char* params = plist; }
HGS_FAMILY(KRow, "row_kernel", "R,N,MODE,NS,PREF,SPLIT");
HGS_FAMILY(KCol, "col_kernel", "R,N,MODE");
HGS_FAMILY(KFused, "col_fused_kernel", "R,N,PHASE,STATS,RULE,NRS");
HGS_FAMILY(KTile, "col_tile_kernel", "R,N,PHASE,NR,STATS,EXTRAS,RULE,LISTED");
HGS_FAMILY(KTile2, "col_tile2_kernel", "R,N,PHASE,NR,RULE,PARK,NXF");
HGS_FAMILY(KPresum, "col_presum_kernel", "R,N,NR");
HGS_FAMILY(KBlue, "bluestein_lines", "R,M");
HGS_FAMILY(KCn2fRun, "c_n2f_run", "R,DEG");
HGS_FAMILY(KCf2nRun, "c_f2n_run", "R,DEG");
HGS_FAMILY(KCn2fPix, "c_n2f_partial", "R,DEG");
HGS_FAMILY(KCf2nPix, "c_f2n", "R,DEG");
HGS_FAMILY(KCgemm, "cgemm_streamk", "R,EPI");
#undef HGS_FAMILY

}  // namespace hgs
