#!/bin/bash
# the round's closing evidence from ONE gpurun call: PMC passes of the final build, then the GPU suite and the two bench lines
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/r04_pmc.sh ${1:-r04y} > /dev/null 2>&1
cat gpurun_out/${1:-r04y}/status.txt | grep -v "rc=0" ; echo "pmc passes done"
python3 tools/make_traffic_json.py gpurun_out/${1:-r04y} profiles/${1:-r04y} > gpurun_out/${1:-r04y}/traffic_current.json
cp gpurun_out/${1:-r04y}/traffic_current.json profiles/traffic_current.json     # (on the box: the bench lines below then describe themselves with it)
OUT_TAG=${2:-r04z} bash tools/r04_final_evidence.sh ${2:-r04z}
