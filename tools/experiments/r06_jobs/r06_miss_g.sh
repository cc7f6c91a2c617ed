bash tools/experiments/r06_jobs/r06_miss419.sh; bash tools/experiments/r06_jobs/r06_miss420.sh
