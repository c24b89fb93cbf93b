cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trs; rocprofv3 --kernel-trace --output-format csv -d /tmp/trs -o t -- python /root/repo/tools/trace_shape.py "$@" > /tmp/trs.log 2>&1 || tail -5 /tmp/trs.log
python /root/repo/tools/trace_summ.py /tmp/trs | tail -9
