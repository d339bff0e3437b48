"""DURATION_IS_GREGORIAN in the daemon's zone (interval.go:84-148 build their civil dates with now.Location()): the engine's
calendar (gubernator_amd/csrc/guber_algo.h, here through the host helpers guber_gregorian_*; the kernels run the same code)
against an independent restatement of Go's time.Date / Location.lookup in Python, cross-checked with the interpreter's zoneinfo
wherever a civil time is unambiguous — at non-UTC offsets, across DST changes in both directions, in a zone east of UTC with a
half-hour offset.  No GPU needed."""
import ctypes as C
import datetime as dt
from zoneinfo import ZoneInfo

import pytest

import gubernator_amd as ga

MS = 1_000_000


# ---- Go's time package, restated (time/zoneinfo.go lookup, time/time.go Date) -------------------------------------------------
def lookup(zone, utc_s):
    off0, tr = zone
    off, start, end = off0, -(1 << 62), (tr[0][0] if tr else (1 << 62))
    for k, (w, o) in enumerate(tr):
        if w > utc_s:
            break
        off, start, end = o, w, (tr[k + 1][0] if k + 1 < len(tr) else (1 << 62))
    return off, start, end


def go_date(zone, y, m, d, hh, mm, ss, ns):
    """time.Date(y, m, d, hh, mm, ss, ns, loc).UnixNano() (m may be 13: AddDate's normalisation)"""
    if m > 12:
        y, m = y + 1, m - 12
    civil = int((dt.datetime(y, m, d, hh, mm, ss, tzinfo=dt.timezone.utc)).timestamp())
    off, start, end = lookup(zone, civil)
    if off != 0:
        utc = civil - off
        if utc < start or utc >= end:
            off, _, _ = lookup(zone, utc)
        civil -= off
    return civil * 1_000_000_000 + ns


def go_expiration(zone, now_ns, d):
    off, _, _ = lookup(zone, now_ns // 1_000_000_000)
    loc = dt.datetime.fromtimestamp(now_ns // 1_000_000_000 + off, dt.timezone.utc)      # now.Date(), now.Hour()
    y, m, dd, hh = loc.year, loc.month, loc.day, loc.hour
    if d == 0:
        return ((now_ns // (60 * 10**9)) * 60 * 10**9 + 60 * 10**9 - 1) // MS
    if d == 1:
        return (go_date(zone, y, m, dd, hh, 0, 0, 0) + 3600 * 10**9 - 1) // MS
    if d == 2:
        return go_date(zone, y, m, dd, 23, 59, 59, 999_999_999) // MS
    if d == 4:
        return (go_date(zone, y, m + 1, 1, 0, 0, 0, 0) - 1) // MS
    if d == 5:
        return (go_date(zone, y + 1, 1, 1, 0, 0, 0, 0) - 1) // MS
    raise ValueError


def go_duration(zone, now_ns, d):
    if d < 3:
        return [60000, 3600000, 86400000][d]
    off, _, _ = lookup(zone, now_ns // 1_000_000_000)
    loc = dt.datetime.fromtimestamp(now_ns // 1_000_000_000 + off, dt.timezone.utc)
    y, m = loc.year, loc.month
    if d == 4:
        begin, end = go_date(zone, y, m, 1, 0, 0, 0, 0), go_date(zone, y, m + 1, 1, 0, 0, 0, 0) - 1
    else:
        begin, end = go_date(zone, y, 1, 1, 0, 0, 0, 0), go_date(zone, y + 1, 1, 1, 0, 0, 0, 0) - 1
    q = abs(begin) // MS * (1 if begin >= 0 else -1)               # Go's / truncates toward zero
    return end - q                                                  # interval.go:99,106: end.UnixNano() - begin.UnixNano()/1000000


def helper(now_ns, d):
    L = ga.lib()
    L.guber_gregorian_expiration.argtypes = [C.c_int64, C.c_int64, C.POINTER(C.c_int64)]
    L.guber_gregorian_duration.argtypes = [C.c_int64, C.c_int64, C.POINTER(C.c_int64)]
    e, g = C.c_int64(0), C.c_int64(0)
    assert L.guber_gregorian_expiration(now_ns, d, C.byref(e)) == 0 and L.guber_gregorian_duration(now_ns, d, C.byref(g)) == 0
    return e.value, g.value


@pytest.fixture(autouse=True)
def back_to_utc():
    yield
    ga.set_timezone()


ZONES = ["America/New_York", "Europe/Berlin", "Asia/Kolkata", "Australia/Lord_Howe"]


@pytest.mark.parametrize("name", ZONES)
def test_calendar_intervals_in_the_daemons_zone(name):
    zone = ga.zone_transitions(name, 2018, 2021)
    assert len(zone[1]) <= 16
    ga.set_timezone(*zone)
    z = ZoneInfo(name)
    # instants: ordinary days, the hours around every transition of 2019 (before, inside the gap / overlap, after), year / month ends
    instants = [dt.datetime(2019, 1, 15, 11, 20, 10, tzinfo=dt.timezone.utc), dt.datetime(2019, 7, 4, 23, 59, 59, tzinfo=dt.timezone.utc),
                dt.datetime(2019, 12, 31, 23, 30, tzinfo=dt.timezone.utc), dt.datetime(2020, 2, 29, 12, 0, tzinfo=dt.timezone.utc)]
    for w, _ in zone[1]:
        for delta in (-7200, -3601, -1800, -1, 0, 1, 1800, 3599, 3600, 7200, 86400):
            instants.append(dt.datetime.fromtimestamp(w + delta, dt.timezone.utc))
    checked = 0
    for t in instants:
        now_ns = int(t.timestamp()) * 1_000_000_000 + 123_456_789
        for d in (0, 1, 2, 4, 5):
            e, g = helper(now_ns, d)
            assert e == go_expiration(zone, now_ns, d), (name, t, d)
            assert g == go_duration(zone, now_ns, d), (name, t, d)
            checked += 1
        # zoneinfo's own answer where it is unambiguous: the end of the local day and of the local month
        loc = t.astimezone(z)
        eod = dt.datetime(loc.year, loc.month, loc.day, 23, 59, 59, tzinfo=z)
        if eod.replace(fold=0).utcoffset() == eod.replace(fold=1).utcoffset():
            assert helper(now_ns, 2)[0] == int(eod.timestamp()) * 1000 + 999, (name, t)
        nm = dt.datetime(loc.year + (loc.month == 12), loc.month % 12 + 1, 1, tzinfo=z)
        assert helper(now_ns, 4)[0] == int(nm.timestamp()) * 1000 - 1, (name, t)
    assert checked >= (100 if zone[1] else 20)


def test_utc_is_the_default_and_a_bad_table_is_refused():
    now_ns = 1_546_341_610 * 1_000_000_000                            # 2019-01-01 11:20:10 UTC (interval_test.go)
    ga.set_timezone()
    assert helper(now_ns, 0)[0] == 1_546_341_659_999 and helper(now_ns, 2)[0] == 1_546_387_199_999
    with pytest.raises(ga.GuberError):
        ga.set_timezone(0, [(200, 3600), (100, 0)])                  # not ascending
    with pytest.raises(ga.GuberError):
        ga.set_timezone(0, [(k, 0) for k in range(17)])              # more than 16
    assert helper(now_ns, 2)[0] == 1_546_387_199_999                  # (refused tables change nothing)


def test_more_transitions_than_the_table_holds_is_a_clear_error():
    """guber_tz_t takes 16 transitions: a zone with daylight saving time over more than eight years does not fit, and the helper says so
    instead of handing guber_set_timezone a list it rejects with INVALID_ARG"""
    import pytest
    with pytest.raises(ValueError, match="at most 16"):
        ga.zone_transitions("America/New_York", 2000, 2020)
    off0, tr = ga.zone_transitions("America/New_York", 2018, 2025)
    assert len(tr) == 16 and off0 == -5 * 3600
