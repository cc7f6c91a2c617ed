# round 6, soak G: 24 more seeds never run before (401..424), 2304 scenes, on the final kernels
bash tools/fuzz_soak.sh r06_soak_g 401 402 403 404 405 406 407 408 409 410 411 412 413 414 415 416 417 418 419 420 421 422 423 424
