"""The legs of bench.py, one module each: common (constants, HIP events, the gather ceiling, host facts),
dump (validation of the headline's ray dump), hostpath, hbm (S-soup-10M), ao (BASELINE config 5), config2, pt (config 4), cpu
(the compiled reference / the port on the host cores: the only place that touches oracle/).  bench.py keeps the contract, the
headline leg and the JSON line."""
