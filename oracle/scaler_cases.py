"""Scripted gradient-norm sequences for the dynamic loss scaler fixture (oracle/gen_scaler_golden.py runs the REFERENCE's
DynamicLossScaler + the fp16 optimizer's clip arithmetic on them; the tests replay them through oracle/restate.py and through
ofa_step_schedule_scaled).  TEST INFRASTRUCTURE.
raw = ||g|| of the (loss-scaled, summed) gradients as the device sees it: float or "inf" / "nan"; n = sample_size."""
INF, NAN = "inf", "nan"
CASES = {
    # default-like: tolerance 0 (every overflow halves the scale), window 4 (grows quickly so the fixture sees it)
    "window4": dict(init_scale=128.0, scale_factor=2.0, scale_window=4, tolerance=0.0, threshold=None, min_loss_scale=1e-4, clip=1.0,
                    seq=[(3000.0, 40), (2500.0, 40), (INF, 40), (1800.0, 41), (90.0, 39), (1700.0, 40), (1600.0, 40), (1500.0, 40),
                         (NAN, 40), (INF, 40), (1400.0, 38), (1300.0, 40), (1200.0, 40), (1100.0, 40), (1000.0, 40), (900.0, 40)]),
    # tolerance: a single overflow inside a long clean run does not shrink the scale; threshold floors it
    "tolerant": dict(init_scale=8.0, scale_factor=2.0, scale_window=3, tolerance=0.3, threshold=4.0, min_loss_scale=1e-4, clip=0.0,
                     seq=[(10.0, 7), (11.0, 7), (12.0, 7), (13.0, 7), (INF, 7), (14.0, 7), (INF, 7), (INF, 7), (15.0, 7), (16.0, 7),
                          (INF, 7), (INF, 7), (INF, 7), (17.0, 7)]),
    # the scale collapses to min_loss_scale: the reference raises FloatingPointError and keeps the previous scale
    "collapse": dict(init_scale=4.0, scale_factor=2.0, scale_window=100, tolerance=0.0, threshold=None, min_loss_scale=1.0, clip=1.0,
                     seq=[(5.0, 3), (INF, 3), (INF, 3), (6.0, 3)]),
}


def raw_value(x):
    return float(x) if isinstance(x, str) else x
