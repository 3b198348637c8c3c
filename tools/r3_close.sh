#!/bin/bash
# round 3: closing visit -- the whole GPU suite, then the evidence passes, on the build that is committed
bash tools/r3_final.sh
bash tools/r3_evidence.sh
