mkdir -p gpurun_out/r8a
Q="--no-cpu-baseline --no-roofline --no-fwd-leg --no-extras --steps 40 --warmup 8"
SY11_TUNE_SAVE=gpurun_out/r8a/picks.bin python bench.py $Q > gpurun_out/r8a/a.json 2> gpurun_out/r8a/a.err
python -c "import json;d=json.load(open('gpurun_out/r8a/a.json'));print('atomic, tuner on (picks saved)',d['ms_per_step'])"
SY11_DETERMINISTIC=1 python bench.py $Q > gpurun_out/r8a/b.json 2> gpurun_out/r8a/b.err
python -c "import json;d=json.load(open('gpurun_out/r8a/b.json'));print('ordered, heuristic tiles',d['ms_per_step'])"
SY11_DETERMINISTIC=1 SY11_TUNE_LOAD=gpurun_out/r8a/picks.bin python bench.py $Q > gpurun_out/r8a/c.json 2> gpurun_out/r8a/c.err
python -c "import json;d=json.load(open('gpurun_out/r8a/c.json'));print('ordered, the saved picks',d['ms_per_step'])"
SY11_TUNE=0 python bench.py $Q > gpurun_out/r8a/d.json 2> gpurun_out/r8a/d.err
python -c "import json;d=json.load(open('gpurun_out/r8a/d.json'));print('atomic, heuristic tiles',d['ms_per_step'])"
