#!/bin/bash
# compile one kernel source with resource-usage remarks and keep the .s under /tmp/cc_check/ (diagnostic helper)
# usage: tools/cc_check.sh wgrad9 [grep-pattern-for-kernel-names]
NAME=$1; PAT=${2:-.}
mkdir -p /tmp/cc_check
cd /root/repo/lstm_ctc_ocr_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Rpass-analysis=kernel-resource-usage -save-temps=obj -c $NAME.hip -o /tmp/cc_check/$NAME.o 2>&1 \
  | grep -E "error|Function Name|VGPRs:|ScratchSize" | sed -e 's/remark: [^ ]* *//' -e 's/\[-Rpass.*//' | paste - - - 2>/dev/null | grep -E "$PAT|error"
