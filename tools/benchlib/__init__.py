"""Support code of /bench.py (the driver contract lives there): the rank launcher, the in-run measurements, the exchange calibration,
the CPU baseline legs, the LongCat workload and the short `also` windows of BASELINE configs 3 / 4.  Lab tooling, not product."""
