// guber_algo.h — per-bucket state machine of the rate-limit path, written once and compiled for the
// gfx950 kernels (hipcc) and, for CPU unit tests of the kernel logic only, for the host (g++).
//
// What it computes follows the reference's algorithms.go (mailgun/gubernator v2):
//   apply()  = one GetRateLimit on one bucket: LRUCache.GetItem expiry check (lrucache.go:111-128,
//              cache.go:43-57) + tokenBucket / leakyBucket (algorithms.go:37-493) + the algorithm
//              switch of handleGetRateLimit (workers.go:293-324).
//   skip()   = k further IDENTICAL requests applied to a bucket, in O(1) for the regimes hot keys
//              live in (plain subtraction, fixed points, period-2 cycles) and by stepping otherwise.
//              This is what lets a batch with thousands of hits on one key be evaluated by
//              independent threads, each reconstructing "the state just before my request".
// How it is organised (one 64-byte record per bucket, explicit `now`, per-request created_at, dense
// flags instead of Go interfaces) is this engine's own design.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GB_HD __host__ __device__ __forceinline__
#else
#define GB_HD inline
#endif

namespace guber {

// ---- enums (gubernator.proto:56-135) --------------------------------------------------------
enum : uint32_t { ALGO_TOKEN = 0, ALGO_LEAKY = 1 };
enum : uint32_t { ST_UNDER = 0, ST_OVER = 1 };
enum : uint32_t {
    BH_NO_BATCHING = 1, BH_GLOBAL = 2, BH_GREGORIAN = 4, BH_RESET_REMAINING = 8, BH_MULTI_REGION = 16,
    BH_DRAIN_OVER_LIMIT = 32
};
enum : uint8_t { IE_OK = 0, IE_INVALID_ALGORITHM = 1, IE_GREG_WEEKS = 2, IE_GREG_INVALID = 3, IE_EMPTY_KEY = 4, IE_RETRY = 5 };

// record kinds: the dynamic type of CacheItem.Value (cache.go:32)
enum : uint32_t { K_ABSENT = 0, K_TOKEN = 1, K_LEAKY = 2, K_NIL = 3 };

// apply() event flags
// EV_ONCHANGE / EV_REMOVE: the persistent Store's callbacks that the reference would issue for this request
// (store.go:49-65): Store.OnChange(r, item) with the item as it is AFTER the request — deferred at
// algorithms.go:149-153 / :382-386 for a found item, direct at :252-254 / :488-490 for a new one, owner only —
// and Store.Remove(key) at :79-84 (token RESET_REMAINING), :96-100 / :311-315 (algorithm switched).
enum : uint32_t { EV_HIT = 1, EV_MISS = 2, EV_OVER = 4, EV_ONCHANGE = 8, EV_REMOVE = 16 };

// ---- Go integer / float semantics -----------------------------------------------------------
GB_HD int64_t wadd(int64_t a, int64_t b) { return (int64_t)((uint64_t)a + (uint64_t)b); }
GB_HD int64_t wsub(int64_t a, int64_t b) { return (int64_t)((uint64_t)a - (uint64_t)b); }
GB_HD int64_t wmul(int64_t a, int64_t b) { return (int64_t)((uint64_t)a * (uint64_t)b); }
// float64 -> int64 as Go on amd64 (CVTTSD2SQ): NaN, +-Inf and out-of-range give INT64_MIN.
GB_HD int64_t go_f2i(double d) {
    if (!(d >= -9223372036854775808.0 && d < 9223372036854775808.0)) return INT64_MIN;
    return (int64_t)d;
}
GB_HD double bits2f(int64_t b) { union { int64_t i; double d; } u; u.i = b; return u.d; }
GB_HD int64_t f2bits(double d) { union { int64_t i; double d; } u; u.d = d; return u.i; }

// ---- one bucket: CacheItem + TokenBucketItem | LeakyBucketItem (cache.go:29-41, store.go:29-43)
struct alignas(16) Rec {
    int64_t limit;       // Limit
    int64_t duration;    // Duration
    int64_t remaining;   // token: Remaining (int64); leaky: Remaining (float64 bits)
    int64_t stamp;       // token: CreatedAt; leaky: UpdatedAt
    int64_t burst;       // leaky: Burst
    int64_t expire_at;   // CacheItem.ExpireAt
    int64_t invalid_at;  // CacheItem.InvalidAt
    uint32_t meta;       // kind (2 bits) | status << 2 (1 bit) | CacheItem.Algorithm << 3 (8 bits) | bits 32..52 of the recency stamp << 11
    uint32_t pad;        // bits 0..31 of the recency stamp
};
// The recency stamp (53 bits): the engine's request sequence number of the LAST request that touched the item (a batch of n
// requests takes n numbers, the i-th request the i-th; Add and GetItem take theirs) — the total order of LRUCache's list
// (lrucache.go:88-128: Add / GetItem move to the front): older stamp = nearer the back.  Unique per item.
constexpr uint32_t REC_META_MASK = 0x7ffu;
constexpr uint64_t REC_STAMP_MASK = (1ull << 53) - 1ull;
static_assert(sizeof(Rec) == 64, "one record = one 64-byte sector");

GB_HD uint32_t rec_kind(const Rec& s) { return s.meta & 3u; }
GB_HD uint32_t rec_status(const Rec& s) { return (s.meta >> 2) & 1u; }
GB_HD uint32_t rec_algo(const Rec& s) { return (s.meta >> 3) & 0xffu; }
GB_HD uint32_t rec_meta(const Rec& s) { return s.meta & REC_META_MASK; }
GB_HD uint32_t make_meta(uint32_t kind, uint32_t status, uint32_t algo) { return (kind & 3u) | ((status & 1u) << 2) | ((algo & 0xffu) << 3); }
GB_HD void rec_set_status(Rec& s, uint32_t st) { s.meta = (s.meta & ~4u) | ((st & 1u) << 2); }
GB_HD uint64_t rec_stamp(const Rec& s) { return ((uint64_t)(s.meta >> 11) << 32) | s.pad; }
GB_HD void rec_set_stamp(Rec& s, uint64_t st) { s.pad = (uint32_t)st; s.meta = (s.meta & REC_META_MASK) | ((uint32_t)((st & REC_STAMP_MASK) >> 32) << 11); }
GB_HD void rec_clear(Rec& s) {
    s.limit = s.duration = s.remaining = s.stamp = s.burst = s.expire_at = s.invalid_at = 0;
    s.meta = 0; s.pad = 0;
}
GB_HD bool rec_eq(const Rec& a, const Rec& b) {
    return a.limit == b.limit && a.duration == b.duration && a.remaining == b.remaining && a.stamp == b.stamp &&
           a.burst == b.burst && a.expire_at == b.expire_at && a.invalid_at == b.invalid_at && rec_meta(a) == rec_meta(b);
}
// cache.go:43-57 IsExpired
GB_HD bool rec_expired(const Rec& s, int64_t now) {
    return (s.invalid_at != 0 && s.invalid_at < now) || s.expire_at < now;
}

struct Req {
    int64_t hits, limit, duration, burst, created_at;
    int64_t greg_expire, greg_duration;  // host-precomputed interval.go values; greg_duration < 0 = -error
    uint32_t behavior;
    uint8_t algorithm, is_owner;
};
GB_HD bool req_eq(const Req& a, const Req& b) {
    return a.hits == b.hits && a.limit == b.limit && a.duration == b.duration && a.burst == b.burst &&
           a.created_at == b.created_at && a.greg_expire == b.greg_expire && a.greg_duration == b.greg_duration &&
           a.behavior == b.behavior && a.algorithm == b.algorithm && a.is_owner == b.is_owner;
}

// Requests that differ ONLY in created_at (RPCs of one device batch stamped a millisecond apart).
GB_HD bool req_eq_but_created(const Req& a, const Req& b) {
    return a.hits == b.hits && a.limit == b.limit && a.duration == b.duration && a.burst == b.burst &&
           a.greg_expire == b.greg_expire && a.greg_duration == b.greg_duration &&
           a.behavior == b.behavior && a.algorithm == b.algorithm && a.is_owner == b.is_owner;
}
// tokenBucket reads r.CreatedAt only when it creates an item (algorithms.go:207) or when the duration
// changed (algorithms.go:135): for a live token bucket with the same duration and no RESET_REMAINING a
// run of requests that differ only in created_at behaves exactly like a run of identical requests.
GB_HD bool created_at_irrelevant(const Rec& s0, const Req& r, int64_t now) {
    return r.algorithm == ALGO_TOKEN && rec_kind(s0) == K_TOKEN && !rec_expired(s0, now) &&
           !(r.behavior & BH_RESET_REMAINING) && s0.duration == r.duration;
}

// leakyBucket reads r.CreatedAt in three places: the leak since UpdatedAt (algorithms.go:361-367), UpdateExpiration
// (:356-358) and the response's ResetTime.  For a LIVE leaky bucket and a request that neither resets nor uses the
// calendar, a request whose created_at lies less than one token interval after UpdatedAt leaks nothing
// (`int64(leak) > 0` is false), so Remaining / UpdatedAt evolve exactly as for any other such created_at; the
// expiration is "the last hit wins" (each request writes created_at + duration itself) and ResetTime is computed
// from the request's own created_at.  If every request of a run on one key is harmless in this sense, the run
// behaves like a run of identical requests as far as the bucket is concerned, and each request can evaluate
// itself with its own created_at.  The second guard keeps UpdateExpiration from making the bucket look expired to
// the requests that follow in the same batch.
GB_HD bool leaky_created_harmless(const Rec& s0, const Req& r, int64_t now) {
    if (r.algorithm != ALGO_LEAKY || rec_kind(s0) != K_LEAKY || rec_expired(s0, now)) return false;
    if ((r.behavior & (BH_RESET_REMAINING | BH_GREGORIAN)) || r.limit <= 0 || r.duration <= 0) return false;
    const double rate = (double)r.duration / (double)r.limit;
    const double leak = (double)wsub(r.created_at, s0.stamp) / rate;
    if (go_f2i(leak) > 0) return false;
    return wadd(r.created_at, r.duration) >= now;
}

// ---- DURATION_IS_GREGORIAN (interval.go:84-148), UTC: the end of the calendar interval `now` falls into, and the
// interval's length.  Proleptic Gregorian calendar from day numbers (no table, no branches on the year): the kernels
// evaluate it per request from the batch clock, the host helpers (guber_gregorian_*) call the same code.
// Wrap-around arithmetic: a clock within one interval of the int64 range must not be undefined behaviour.
// Returns 0 or the reference's error as an item code (IE_GREG_WEEKS / IE_GREG_INVALID).
namespace cal {
constexpr int64_t kSec = 1000000000LL, kMin = 60 * kSec, kHour = 3600 * kSec, kDay = 86400 * kSec, kMs = 1000000LL;
GB_HD int64_t floor_div(int64_t a, int64_t b) { const int64_t q = a / b; return (a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q; }
GB_HD int64_t days_from_civil(int64_t y, int m, int d) {
    y -= m <= 2;
    const int64_t era = floor_div(y, 400);
    const int64_t yoe = y - era * 400;
    const int64_t doy = (153 * (m + (m > 2 ? -3 : 9)) + 2) / 5 + d - 1;
    const int64_t doe = yoe * 365 + yoe / 4 - yoe / 100 + doy;
    return era * 146097 + doe - 719468;
}
GB_HD void civil_from_days(int64_t z, int64_t& y, int& m, int& d) {
    z += 719468;
    const int64_t era = floor_div(z, 146097);
    const int64_t doe = z - era * 146097;
    const int64_t yoe = (doe - doe / 1460 + doe / 36524 - doe / 146096) / 365;
    const int64_t doy = doe - (365 * yoe + yoe / 4 - yoe / 100);
    const int64_t mp = (5 * doy + 2) / 153;
    d = (int)(doy - (153 * mp + 2) / 5 + 1);
    m = (int)(mp < 10 ? mp + 3 : mp - 9);
    y = yoe + era * 400 + (m <= 2);
}
}  // namespace cal
// ---- the daemon's time zone (interval.go:97-142 build their civil dates with now.Location()) --------------------------------
// A zone = the UTC offset in effect before the first listed transition and, per transition, the instant (UTC seconds) and the
// offset from then on — the shape of Go's time.Location (zoneTrans / zone).  n == 0 and offset0 == 0 is UTC.  The engine keeps
// one per process (guber_set_timezone), the kernels read it from a device global.
constexpr int TZ_MAX = 16;
struct TzTable { uint32_t n; int32_t offset0_s; int64_t when_s[TZ_MAX]; int32_t offset_s[TZ_MAX]; };
namespace cal {
// Location.lookup: the period [start, end) containing the instant, and its offset
GB_HD void tz_lookup(const TzTable* tz, int64_t utc_s, int64_t& off, int64_t& start, int64_t& end) {
    off = 0; start = INT64_MIN; end = INT64_MAX;
    if (!tz) return;
    off = tz->offset0_s;
    if (tz->n) end = tz->when_s[0];
    for (uint32_t k = 0; k < tz->n && k < (uint32_t)TZ_MAX; ++k) {
        if (tz->when_s[k] > utc_s) break;
        off = tz->offset_s[k]; start = tz->when_s[k]; end = k + 1 < tz->n ? tz->when_s[k + 1] : INT64_MAX;
    }
}
// time.Date(..., loc): civil seconds of the zone -> UTC seconds, with Go's two-step resolution of the offset (time.go Date():
// look the civil value up as if it were UTC; if the resulting instant falls outside that period, look the instant itself up)
GB_HD int64_t tz_civil_to_utc_s(const TzTable* tz, int64_t civil_s) {
    int64_t off, start, end;
    tz_lookup(tz, civil_s, off, start, end);
    if (off != 0) {
        const int64_t utc = civil_s - off;
        if (utc < start || utc >= end) { int64_t s2, e2; tz_lookup(tz, utc, off, s2, e2); }
        return civil_s - off;
    }
    return civil_s;
}
}  // namespace cal
// interval.go:117-148 GregorianExpiration
GB_HD uint32_t greg_expiration(int64_t now_ns, int64_t d, int64_t& expire_ms, const TzTable* tz = nullptr) {
    using namespace cal;
    expire_ms = 0;
    int64_t y; int m, dd;
    int64_t off = 0, ps, pe;
    tz_lookup(tz, floor_div(now_ns, kSec), off, ps, pe);
    const int64_t civil_ns = wadd(now_ns, wmul(off, kSec));                        // now.Date(), now.Hour(): the zone's civil time
    const int64_t day = floor_div(civil_ns, kDay);
    civil_from_days(day, y, m, dd);
    int64_t end_ns;
    if (d == 0) end_ns = wadd(wmul(floor_div(now_ns, kMin), kMin), kMin - 1);      // now.Truncate(Minute): on the absolute time
    else if (d == 1) {
        if (!tz) end_ns = wadd(wmul(floor_div(now_ns, kHour), kHour), kHour - 1);
        else {
            const int64_t hour = floor_div(civil_ns - day * kDay, kHour);
            end_ns = wadd(wmul(tz_civil_to_utc_s(tz, day * 86400 + hour * 3600), kSec), kHour - 1);
        }
    }
    else if (d == 2) end_ns = tz ? wadd(wmul(tz_civil_to_utc_s(tz, day * 86400 + 86399), kSec), kSec - 1) : wadd(wmul(day, kDay), kDay - 1);
    else if (d == 3) return 2u;                                                     // IE_GREG_WEEKS
    else if (d == 4) {
        const int64_t nd = days_from_civil(m == 12 ? y + 1 : y, m == 12 ? 1 : m + 1, 1);
        end_ns = tz ? wsub(wmul(tz_civil_to_utc_s(tz, nd * 86400), kSec), 1) : wsub(wmul(nd, kDay), 1);
    }
    else if (d == 5) {
        const int64_t nd = days_from_civil(y + 1, 1, 1);
        end_ns = tz ? wsub(wmul(tz_civil_to_utc_s(tz, nd * 86400), kSec), 1) : wsub(wmul(nd, kDay), 1);
    }
    else return 3u;                                                                 // IE_GREG_INVALID
    expire_ms = floor_div(end_ns, kMs);
    return 0u;
}
// interval.go:84-110 GregorianDuration.  The months / years arms keep the reference's expression exactly as written:
//   end.UnixNano() - begin.UnixNano()/1000000
GB_HD uint32_t greg_duration(int64_t now_ns, int64_t d, int64_t& duration, const TzTable* tz = nullptr) {
    using namespace cal;
    duration = 0;
    if (d == 0) { duration = 60000; return 0u; }
    if (d == 1) { duration = 3600000; return 0u; }
    if (d == 2) { duration = 86400000; return 0u; }
    if (d == 3) return 2u;
    if (d != 4 && d != 5) return 3u;
    int64_t y; int m, dd;
    int64_t off = 0, ps, pe;
    tz_lookup(tz, floor_div(now_ns, kSec), off, ps, pe);
    civil_from_days(floor_div(wadd(now_ns, wmul(off, kSec)), kDay), y, m, dd);
    int64_t begin, end;
    const int64_t d0 = d == 4 ? days_from_civil(y, m, 1) : days_from_civil(y, 1, 1);
    const int64_t d1 = d == 4 ? days_from_civil(m == 12 ? y + 1 : y, m == 12 ? 1 : m + 1, 1) : days_from_civil(y + 1, 1, 1);
    if (tz) {
        begin = wmul(tz_civil_to_utc_s(tz, d0 * 86400), kSec);
        end = wsub(wmul(tz_civil_to_utc_s(tz, d1 * 86400), kSec), 1);
    } else {
        begin = wmul(d0, kDay);
        end = wsub(wmul(d1, kDay), 1);
    }
    duration = wsub(end, begin / kMs);
    return 0u;
}
// the two values a request carries when the host has not precomputed them: greg_duration < 0 = -error, as in guber_batch_t
GB_HD void greg_fill(int64_t now_ms, int64_t d, int64_t& g_expire, int64_t& g_duration, const TzTable* tz = nullptr) {
    const int64_t now_ns = wmul(now_ms, cal::kMs);
    int64_t e = 0, du = 0;
    uint32_t rc = greg_expiration(now_ns, d, e, tz);
    if (rc == 0u) rc = greg_duration(now_ns, d, du, tz);
    g_expire = e; g_duration = rc ? -(int64_t)rc : du;
}

struct Resp {
    int64_t limit, remaining, reset_time;
    uint8_t status, err;
};
GB_HD void resp_clear(Resp& r) { r.limit = r.remaining = r.reset_time = 0; r.status = 0; r.err = 0; }

// algorithms.go:206-257 tokenBucketNewItem
GB_HD uint32_t token_new_item(Rec& s, const Req& r, Resp& rl) {
    uint32_t ev = 0;
    int64_t expire = wadd(r.created_at, r.duration);
    int64_t remaining = wsub(r.limit, r.hits);
    if (r.behavior & BH_GREGORIAN) {
        if (r.greg_duration < 0) { resp_clear(rl); rl.err = (uint8_t)(-r.greg_duration); return ev; }
        expire = r.greg_expire;
    }
    rl.status = ST_UNDER; rl.limit = r.limit; rl.remaining = remaining; rl.reset_time = expire;
    if (r.hits > r.limit) {
        if (r.is_owner) ev |= EV_OVER;
        rl.status = ST_OVER; rl.remaining = r.limit; remaining = r.limit;
    }
    rec_clear(s);
    s.limit = r.limit; s.duration = r.duration; s.remaining = remaining; s.stamp = r.created_at;
    s.expire_at = expire; s.meta = make_meta(K_TOKEN, ST_UNDER, ALGO_TOKEN);
    return ev;
}

// algorithms.go:437-493 leakyBucketNewItem (`burst` = r.Burst after the :264 defaulting)
GB_HD uint32_t leaky_new_item(Rec& s, const Req& r, int64_t burst, int64_t now, Resp& rl) {
    uint32_t ev = 0;
    int64_t duration = r.duration;
    double rate = (double)duration / (double)r.limit;
    if (r.behavior & BH_GREGORIAN) {
        if (r.greg_duration < 0) { resp_clear(rl); rl.err = (uint8_t)(-r.greg_duration); return ev; }
        duration = wsub(r.greg_expire, now);
    }
    int64_t bh = wsub(burst, r.hits);
    double remaining = (double)bh;
    int64_t irate = go_f2i(rate);
    rl.status = ST_UNDER; rl.limit = r.limit; rl.remaining = bh;
    rl.reset_time = wadd(r.created_at, wmul(wsub(r.limit, bh), irate));
    if (r.hits > burst) {
        if (r.is_owner) ev |= EV_OVER;
        rl.status = ST_OVER; rl.remaining = 0;
        rl.reset_time = wadd(r.created_at, wmul(wsub(rl.limit, rl.remaining), irate));
        remaining = 0.0;
    }
    rec_clear(s);
    s.limit = r.limit; s.duration = duration; s.remaining = f2bits(remaining); s.stamp = r.created_at;
    s.burst = burst; s.expire_at = wadd(r.created_at, duration);
    s.meta = make_meta(K_LEAKY, 0, r.algorithm);
    return ev;
}

// One GetRateLimit on one bucket.  `s` is the bucket before and after; returns EV_* flags.
GB_HD uint32_t apply(Rec& s, const Req& r, int64_t now, Resp& rl) {
    resp_clear(rl);
    if (r.algorithm > ALGO_LEAKY) { rl.err = IE_INVALID_ALGORITHM; return 0; }  // workers.go:317-321
    uint32_t ev;
    bool ok = false;
    // lrucache.go:111-128 GetItem: an expired item is removed and reported as a miss
    if (rec_kind(s) != K_ABSENT) {
        if (rec_expired(s, now)) { rec_clear(s); ev = EV_MISS; }
        else { ev = EV_HIT; ok = true; }
    } else ev = EV_MISS;
    if (ok && rec_kind(s) == K_NIL) ok = false;  // algorithms.go:55-63 / :284-292 "Value is nil"

    // Store.OnChange after a NewItem (:252-254, :488-490): owner only, not on the Gregorian error return
    const uint32_t chg = r.is_owner ? EV_ONCHANGE : 0u;
    if (r.algorithm == ALGO_TOKEN) {
        if (!ok) { ev |= token_new_item(s, r, rl); return rl.err ? ev : ev | chg; }          // :202
        if (r.behavior & BH_RESET_REMAINING) {                               // :78-90
            rec_clear(s);
            rl.status = ST_UNDER; rl.limit = r.limit; rl.remaining = r.limit; rl.reset_time = 0;
            return ev | EV_REMOVE;
        }
        if (rec_kind(s) != K_TOKEN) {                                        // :91-103
            rec_clear(s);
            ev |= EV_REMOVE | token_new_item(s, r, rl);
            return rl.err ? ev : ev | chg;
        }
        if (s.limit != r.limit) {                                            // :106-113
            s.remaining = wadd(s.remaining, wsub(r.limit, s.limit));
            if (s.remaining < 0) s.remaining = 0;
            s.limit = r.limit;
        }
        rl.status = (uint8_t)rec_status(s); rl.limit = r.limit;              // :115-120
        rl.remaining = s.remaining; rl.reset_time = s.expire_at;
        if (s.duration != r.duration) {                                      // :123-147
            int64_t expire = wadd(s.stamp, r.duration);
            if (r.behavior & BH_GREGORIAN) {
                if (r.greg_duration < 0) { resp_clear(rl); rl.err = (uint8_t)(-r.greg_duration); return ev; }
                expire = r.greg_expire;
            }
            if (expire <= r.created_at) {                                    // renew; rl.remaining stays stale
                expire = wadd(r.created_at, r.duration);
                s.stamp = r.created_at;
                s.remaining = s.limit;
            }
            s.expire_at = expire; s.duration = r.duration; rl.reset_time = expire;
        }
        ev |= chg;                                                           // :149-153 deferred OnChange: every return below
        if (r.hits == 0) return ev;                                          // :157-159
        if (rl.remaining == 0 && r.hits > 0) {                               // :162-170 sticky status
            if (r.is_owner) ev |= EV_OVER;
            rl.status = ST_OVER; rec_set_status(s, ST_OVER);
            return ev;
        }
        if (s.remaining == r.hits) { s.remaining = 0; rl.remaining = 0; return ev; }  // :173-178
        if (r.hits > s.remaining) {                                          // :182-194
            if (r.is_owner) ev |= EV_OVER;
            rl.status = ST_OVER;
            if (r.behavior & BH_DRAIN_OVER_LIMIT) { s.remaining = 0; rl.remaining = 0; }
            return ev;
        }
        s.remaining = wsub(s.remaining, r.hits); rl.remaining = s.remaining;  // :196-198
        return ev;
    }

    // ---- leaky bucket, algorithms.go:260-434
    int64_t burst = r.burst;
    if (burst == 0) burst = r.limit;                                         // :264-266
    if (!ok) { ev |= leaky_new_item(s, r, burst, now, rl); return rl.err ? ev : ev | chg; }   // :433
    if (rec_kind(s) != K_LEAKY) {                                            // :308-318
        rec_clear(s);
        ev |= EV_REMOVE | leaky_new_item(s, r, burst, now, rl);
        return rl.err ? ev : ev | chg;
    }
    double rem = bits2f(s.remaining);
    if (r.behavior & BH_RESET_REMAINING) rem = (double)burst;                // :320-322
    if (s.burst != burst) {                                                  // :325-330
        if (burst > go_f2i(rem)) rem = (double)burst;
        s.burst = burst;
    }
    s.limit = r.limit; s.duration = r.duration;                              // :332-333
    int64_t duration = r.duration;
    double rate = (double)duration / (double)r.limit;                        // :336
    if (r.behavior & BH_GREGORIAN) {                                         // :338-354
        if (r.greg_duration < 0) { s.remaining = f2bits(rem); resp_clear(rl); rl.err = (uint8_t)(-r.greg_duration); return ev; }
        rate = (double)r.greg_duration / (double)r.limit;
        duration = wsub(r.greg_expire, now);
    }
    ev |= chg;                                                               // :382-386 deferred OnChange (the returns above it are errors)
    if (r.hits != 0) s.expire_at = wadd(r.created_at, duration);             // :356-358 UpdateExpiration
    int64_t elapsed = wsub(r.created_at, s.stamp);                           // :361-367
    double leak = (double)elapsed / rate;
    if (go_f2i(leak) > 0) { rem = rem + leak; s.stamp = r.created_at; }
    if (go_f2i(rem) > s.burst) rem = (double)s.burst;                        // :369-371
    int64_t irate = go_f2i(rate);
    int64_t irem = go_f2i(rem);
    rl.limit = s.limit; rl.remaining = irem; rl.status = ST_UNDER;           // :373-378
    rl.reset_time = wadd(r.created_at, wmul(wsub(s.limit, irem), irate));
    if (irem == 0 && r.hits > 0) {                                           // :389-395
        if (r.is_owner) ev |= EV_OVER;
        rl.status = ST_OVER;
    } else if (irem == r.hits) {                                             // :398-403
        rem = 0.0; rl.remaining = 0;
        rl.reset_time = wadd(r.created_at, wmul(wsub(rl.limit, rl.remaining), irate));
    } else if (r.hits > irem) {                                              // :407-420
        if (r.is_owner) ev |= EV_OVER;
        rl.status = ST_OVER;
        if (r.behavior & BH_DRAIN_OVER_LIMIT) { rem = 0.0; rl.remaining = 0; }
    } else if (r.hits != 0) {                                                // :423-430
        rem = rem - (double)r.hits;
        rl.remaining = go_f2i(rem);
        rl.reset_time = wadd(r.created_at, wmul(wsub(rl.limit, rl.remaining), irate));
    }
    s.remaining = f2bits(rem);
    return ev;
}

// True when `after` is `before` with Remaining reduced by exactly r.hits (> 0) and nothing else
// touched: the "plain subtraction" step of algorithms.go:196-198 / :427-430.
// The guards make the extrapolation in skip() safe: the bucket is live at `now`, the request does
// not reset or reconfigure it, so the next step sees the same branch conditions except Remaining.
GB_HD bool pure_subtract(const Rec& before, const Rec& after, const Req& r, int64_t now) {
    if (r.hits <= 0 || (r.behavior & BH_RESET_REMAINING)) return false;
    if (before.limit != after.limit || before.duration != after.duration || before.stamp != after.stamp ||
        before.burst != after.burst || before.expire_at != after.expire_at || before.invalid_at != after.invalid_at ||
        rec_meta(before) != rec_meta(after))
        return false;
    if (rec_expired(after, now) || after.limit != r.limit) return false;
    uint32_t k = rec_kind(after);
    if (k == K_TOKEN) {
        if (r.algorithm != ALGO_TOKEN || after.duration != r.duration) return false;
        return before.remaining > 0 && after.remaining == before.remaining - r.hits && after.remaining >= 0;
    }
    if (k == K_LEAKY) {
        if (r.algorithm != ALGO_LEAKY || after.burst != (r.burst == 0 ? r.limit : r.burst)) return false;
        double b = bits2f(before.remaining), a = bits2f(after.remaining);
        return b >= 0.0 && b < 9007199254740992.0 && a >= 0.0 && a == b - (double)r.hits;
    }
    return false;
}

// Advance `s` by k further applications of the SAME request r (same now), exactly.
//  * fixed point (apply leaves the bucket unchanged): the remaining steps are identity;
//  * period-2 cycle (e.g. token RESET_REMAINING alternating remove / create): parity decides;
//  * an observed plain subtraction is extrapolated: with n = int(Remaining) after the step, the next
//    (n-1)/hits steps are plain subtractions too (a step is plain iff hits < n); for the leaky
//    bucket every intermediate float64 is exact because hits is an integer, Remaining < 2^53 and the
//    magnitude only shrinks, so R - j*hits in one operation equals j successive subtractions bit for bit;
//  * anything else is stepped one request at a time (always correct, O(k)).
GB_HD void skip(Rec& s, const Req& r, int64_t now, uint64_t k) {
    Rec prev2; rec_clear(prev2);
    bool have_prev2 = false;
    Resp tmp;
    while (k > 0) {
        Rec before = s;
        apply(s, r, now, tmp);
        k--;
        if (rec_eq(s, before)) return;                      // fixed point
        if (have_prev2 && rec_eq(s, prev2)) {               // period 2: s == state two steps ago
            if (k & 1) s = before;
            return;
        }
        prev2 = before; have_prev2 = true;
        if (k > 0 && pure_subtract(before, s, r, now)) {
            uint32_t kind = rec_kind(s);
            int64_t n = kind == K_TOKEN ? s.remaining : go_f2i(bits2f(s.remaining));
            if (n > 0) {
                uint64_t m = (uint64_t)(n - 1) / (uint64_t)r.hits;
                uint64_t j = m < k ? m : k;
                if (j > 0) {
                    int64_t dec = (int64_t)(j * (uint64_t)r.hits);  // <= n-1, exact
                    if (kind == K_TOKEN) s.remaining -= dec;
                    else s.remaining = f2bits(bits2f(s.remaining) - (double)dec);
                    k -= j;
                    have_prev2 = false;
                }
            }
        }
    }
}

// Response of the request of rank `rank` (0-based) in a run of identical requests applied to the
// bucket s0; `s_after` is the bucket right after that request (the run's final state when rank is the
// last).  Every thread of a run calls this independently: no inter-thread dependency, no atomics.
GB_HD uint32_t eval_uniform_rank(const Rec& s0, const Req& r, int64_t now, uint64_t rank, Resp& out, Rec& s_after) {
    s_after = s0;
    if (rank > 0) {
        Resp tmp;
        apply(s_after, r, now, tmp);
        skip(s_after, r, now, rank - 1);
    }
    return apply(s_after, r, now, out);
}

// eval_uniform_rank with ONE apply() site (the kernels inline it: a third of the code and registers of the three-site
// form above).  k = applications still due before the request's own; the shortcuts of skip() are taken after every
// application, the first one included — each is exact wherever it triggers, so the results are identical.
GB_HD uint32_t eval_uniform_rank_1x(const Rec& s0, const Req& r, int64_t now, uint64_t rank, Resp& out, Rec& s) {
    s = s0;
    uint64_t k = rank;
    Rec prev2; rec_clear(prev2);
    bool have_prev2 = false;
    for (;;) {
        const Rec before = s;
        const uint32_t ev = apply(s, r, now, out);
        if (k == 0) return ev;
        k--;
        if (k == 0) continue;
        if (rec_eq(s, before)) { k = 0; continue; }                       // fixed point
        if (have_prev2 && rec_eq(s, prev2)) {                             // period 2
            if (k & 1) s = before;
            k = 0;
            continue;
        }
        prev2 = before; have_prev2 = true;
        if (pure_subtract(before, s, r, now)) {
            const uint32_t kind = rec_kind(s);
            const int64_t n = kind == K_TOKEN ? s.remaining : go_f2i(bits2f(s.remaining));
            if (n > 0) {
                const uint64_t m = (uint64_t)(n - 1) / (uint64_t)r.hits;
                const uint64_t j = m < k ? m : k;
                if (j > 0) {
                    const int64_t dec = (int64_t)(j * (uint64_t)r.hits);  // <= n-1, exact
                    if (kind == K_TOKEN) s.remaining -= dec;
                    else s.remaining = f2bits(bits2f(s.remaining) - (double)dec);
                    k -= j;
                    have_prev2 = false;
                }
            }
        }
    }
}

// ---- closed forms for the regimes real traffic lives in ---------------------------------------
// A run of identical requests with hits = h > 0 against a LIVE bucket whose configuration the request does not
// change is a counter walk on the integer level I (token: Remaining; leaky: int64(Remaining)): while I > h the
// request subtracts (algorithms.go:196-198 / :427-430); the request that finds I == h takes the last tokens
// (:173-178 / :398-403); one that finds 0 < I < h is refused (:182-194 / :407-420) and leaves the level alone
// unless DRAIN_OVER_LIMIT zeroes it; one that finds I == 0 is refused (:162-170 / :389-395).  So the request of
// rank k can tell which of these it is from (I0, h, k) alone — one 64x64 multiply, no stepping, and a division only
// for the requests that come after a refused one without DRAIN.  token_fast / leaky_fast return exactly what
// eval_uniform_rank returns for the same arguments (tests/test_kernel_logic_host.py fuzzes the equality); everything
// they decline goes through apply() / skip().
enum : uint32_t { RUN_PLAIN = 0, RUN_EXACT = 1, RUN_SHORT = 2, RUN_STUCK = 3, RUN_ZERO = 4 };

// a * h < lim without overflow (a < 2^32, h and lim < 2^63)
GB_HD bool mul_lt(uint64_t a, uint64_t h, uint64_t lim) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, h) == 0ull && a * h < lim;
#else
    return (unsigned __int128)a * h < (unsigned __int128)lim;
#endif
}
// What request k (0-based) of the run finds: `level` = the integer level right before it acts.
GB_HD uint32_t run_position(uint64_t I0, uint64_t h, uint64_t k, bool drain, uint64_t& level) {
    if (I0 == 0) { level = 0; return RUN_ZERO; }
    if (mul_lt(k + 1, h, I0)) { level = I0 - k * h; return RUN_PLAIN; }       // k < P = (I0 - 1) / h
    if (mul_lt(k, h, I0)) { level = I0 - k * h; return level == h ? RUN_EXACT : RUN_SHORT; }   // k == P
    if (drain || h == 1) { level = 0; return RUN_ZERO; }                      // the level was zeroed at request P
    const uint64_t lp = (I0 - 1) % h + 1;                                     // level request P found, 1..h
    if (lp == h) { level = 0; return RUN_ZERO; }
    level = lp;
    return RUN_STUCK;
}

GB_HD bool token_fast_ok(const Rec& s0, const Req& r, int64_t now) {
    return r.algorithm == ALGO_TOKEN && rec_kind(s0) == K_TOKEN && !rec_expired(s0, now) &&
           !(r.behavior & BH_RESET_REMAINING) && s0.limit == r.limit && s0.duration == r.duration && r.hits > 0 &&
           s0.remaining >= 0;
}
GB_HD uint32_t token_fast(const Rec& s0, const Req& r, uint64_t k, Resp& out, Rec& after) {
    uint64_t level;
    const bool drain = (r.behavior & BH_DRAIN_OVER_LIMIT) != 0;
    const uint32_t kind = run_position((uint64_t)s0.remaining, (uint64_t)r.hits, k, drain, level);
    uint32_t ev = EV_HIT | (r.is_owner ? EV_ONCHANGE : 0u);
    after = s0;
    out.err = 0; out.limit = r.limit; out.reset_time = s0.expire_at; out.status = (uint8_t)rec_status(s0);
    if (kind == RUN_PLAIN) { after.remaining = (int64_t)level - r.hits; out.remaining = after.remaining; }
    else if (kind == RUN_EXACT) { after.remaining = 0; out.remaining = 0; }
    else if (kind == RUN_ZERO) {
        after.remaining = 0; out.remaining = 0; out.status = ST_OVER; rec_set_status(after, ST_OVER);
        if (r.is_owner) ev |= EV_OVER;
    } else {                                                                  // RUN_SHORT / RUN_STUCK
        out.status = ST_OVER;
        if (r.is_owner) ev |= EV_OVER;
        after.remaining = drain ? 0 : (int64_t)level; out.remaining = after.remaining;
    }
    return ev;
}

// false = not a case for the closed form (the caller uses eval_uniform_rank)
GB_HD bool leaky_fast(const Rec& s0, const Req& r, int64_t now, uint64_t k, Resp& out, Rec& after, uint32_t& ev_out) {
    if (r.algorithm != ALGO_LEAKY || rec_kind(s0) != K_LEAKY || rec_expired(s0, now)) return false;
    if ((r.behavior & (BH_RESET_REMAINING | BH_GREGORIAN)) || r.hits <= 0) return false;
    const int64_t burst = r.burst == 0 ? r.limit : r.burst;
    if (s0.burst != burst) return false;
    if (wadd(r.created_at, r.duration) < now) return false;   // UpdateExpiration (:356-358) would expire the bucket for the next request of the run
    // what every request of the run does before it looks at the level (algorithms.go:332-378); only the first one
    // can leak — it moves UpdatedAt to created_at, and one that does not leak leaves the same elapsed time behind
    double rem = bits2f(s0.remaining);
    const double rate = (double)r.duration / (double)r.limit;
    const double leak = (double)wsub(r.created_at, s0.stamp) / rate;
    const bool leaked = go_f2i(leak) > 0;
    if (leaked) rem = rem + leak;
    if (go_f2i(rem) > burst) rem = (double)burst;
    if (!(rem >= 0.0 && rem < 9007199254740992.0)) return false;
    const int64_t irate = go_f2i(rate);
    const uint64_t I0 = (uint64_t)go_f2i(rem), h = (uint64_t)r.hits;
    uint64_t level;
    const bool drain = (r.behavior & BH_DRAIN_OVER_LIMIT) != 0;
    const uint32_t kind = run_position(I0, h, k, drain, level);
    uint32_t ev = EV_HIT | (r.is_owner ? EV_ONCHANGE : 0u);
    double rem_a;
    out.err = 0; out.limit = r.limit; out.status = ST_UNDER;
    if (kind == RUN_PLAIN) {
        rem_a = rem - (double)(I0 - level + h);           // (k + 1) * h, exact: an integer below 2^53 off a double below 2^53
        out.remaining = (int64_t)(level - h);
    } else if (kind == RUN_EXACT) {
        rem_a = 0.0; out.remaining = 0;
    } else if (kind == RUN_ZERO) {
        rem_a = I0 == 0 ? rem : 0.0; out.remaining = 0; out.status = ST_OVER;
        if (r.is_owner) ev |= EV_OVER;
    } else {
        out.status = ST_OVER;
        if (r.is_owner) ev |= EV_OVER;
        out.remaining = (int64_t)level;
        rem_a = drain ? 0.0 : rem - (double)(I0 - level);
    }
    out.reset_time = wadd(r.created_at, wmul(wsub(r.limit, out.remaining), irate));
    if ((kind == RUN_SHORT || kind == RUN_STUCK) && drain) out.remaining = 0;   // :417-420: ResetTime is not recomputed
    after = s0;
    after.limit = r.limit; after.duration = r.duration; after.remaining = f2bits(rem_a);
    if (leaked) after.stamp = r.created_at;
    after.expire_at = wadd(r.created_at, r.duration);
    ev_out = ev;
    return true;
}

// Response of the request of rank `rank` in a run of identical requests: closed form where it applies,
// apply() / skip() otherwise.  `generic` is the slow path (the kernels pass an out-of-line function so that the
// common case keeps a small register footprint).
template <typename Generic>
GB_HD uint32_t eval_rank(const Rec& s0, const Req& r, int64_t now, uint64_t rank, Resp& out, Rec& after, Generic generic) {
    if (token_fast_ok(s0, r, now)) return token_fast(s0, r, rank, out, after);
    uint32_t ev;
    if (leaky_fast(s0, r, now, rank, out, after, ev)) return ev;
    return generic(s0, r, now, rank, out, after);
}

// ---- hashes on the path ----------------------------------------------------------------------
// XXH64 (published xxHash spec); the reference hashes the key with it to pick a worker shard
// (workers.go:153-155, OneOfOne/xxhash v1.2.8); here it addresses the HBM bucket directory.
namespace xxh {
constexpr uint64_t P1 = 0x9E3779B185EBCA87ULL, P2 = 0xC2B2AE3D27D4EB4FULL, P3 = 0x165667B19E3779F9ULL,
                   P4 = 0x85EBCA77C2B2AE63ULL, P5 = 0x27D4EB2F165667C5ULL;
GB_HD uint64_t rotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
GB_HD uint64_t round1(uint64_t acc, uint64_t in) { return rotl(acc + in * P2, 31) * P1; }
GB_HD uint64_t merge(uint64_t acc, uint64_t v) { return (acc ^ round1(0, v)) * P1 + P4; }
GB_HD uint64_t ld64(const uint8_t* p) {
    uint64_t v; __builtin_memcpy(&v, p, 8); return v;
}
GB_HD uint32_t ld32(const uint8_t* p) {
    uint32_t v; __builtin_memcpy(&v, p, 4); return v;
}
}  // namespace xxh

GB_HD uint64_t xxhash64(const uint8_t* p, uint32_t len, uint64_t seed) {
    using namespace xxh;
    const uint8_t* end = p + len;
    uint64_t h;
    if (len >= 32) {
        uint64_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
        const uint8_t* lim = end - 32;
        do {
            v1 = round1(v1, ld64(p)); v2 = round1(v2, ld64(p + 8));
            v3 = round1(v3, ld64(p + 16)); v4 = round1(v4, ld64(p + 24));
            p += 32;
        } while (p <= lim);
        h = rotl(v1, 1) + rotl(v2, 7) + rotl(v3, 12) + rotl(v4, 18);
        h = merge(h, v1); h = merge(h, v2); h = merge(h, v3); h = merge(h, v4);
    } else {
        h = seed + P5;
    }
    h += (uint64_t)len;
    while (p + 8 <= end) { h ^= round1(0, ld64(p)); h = rotl(h, 27) * P1 + P4; p += 8; }
    if (p + 4 <= end) { h ^= (uint64_t)ld32(p) * P1; h = rotl(h, 23) * P2 + P3; p += 4; }
    while (p < end) { h ^= (uint64_t)(*p) * P5; h = rotl(h, 11) * P1; p++; }
    h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
    return h;
}

// the same for a key of len < 32 bytes whose first four 8-byte words are already in registers (k_front fetches the key words
// speculatively, before it knows the key's offset for certain): no memory access
GB_HD uint64_t xxhash64_words4(const uint64_t w[4], uint32_t len, uint64_t seed) {
    using namespace xxh;
    uint64_t h = seed + P5 + (uint64_t)len;
    const uint32_t n8 = len >> 3;
#ifdef __clang__
#pragma unroll
#endif
    for (uint32_t k = 0; k < 3; ++k)
        if (k < n8) { h ^= round1(0, w[k]); h = rotl(h, 27) * P1 + P4; }
    uint64_t t = n8 == 0 ? w[0] : n8 == 1 ? w[1] : n8 == 2 ? w[2] : w[3];   // the word holding the tail bytes
    uint32_t rest = len & 7u;
    if (rest >= 4) { h ^= (uint64_t)(uint32_t)t * P1; h = rotl(h, 23) * P2 + P3; t >>= 32; rest -= 4; }
#ifdef __clang__
#pragma unroll
#endif
    for (uint32_t k = 0; k < 3; ++k)
        if (k < rest) { h ^= (t & 0xffull) * P5; h = rotl(h, 11) * P1; t >>= 8; }
    h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
    return h;
}

// FNV-1 / FNV-1a 64 (segmentio/fasthash v1.0.2; replicated_hash.go:33, config.go:429-433)
GB_HD uint64_t fnv1_64(const uint8_t* p, uint32_t len) {
    uint64_t h = 0xcbf29ce484222325ULL;
    for (uint32_t i = 0; i < len; i++) { h *= 0x100000001b3ULL; h ^= p[i]; }
    return h;
}
GB_HD uint64_t fnv1a_64(const uint8_t* p, uint32_t len) {
    uint64_t h = 0xcbf29ce484222325ULL;
    for (uint32_t i = 0; i < len; i++) { h ^= p[i]; h *= 0x100000001b3ULL; }
    return h;
}

}  // namespace guber
