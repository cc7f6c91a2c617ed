# round 6, soak F: 24 more seeds never run before (301..324), 2304 scenes, on the final kernels
bash tools/fuzz_soak.sh r06_soak_f 301 302 303 304 305 306 307 308 309 310 311 312 313 314 315 316 317 318 319 320 321 322 323 324
