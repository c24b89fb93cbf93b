cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/trb -o t -- python /root/repo/tools/trace_batched.py ${1:-8} ${2:-1024} > /tmp/trb.log 2>&1 || tail -5 /tmp/trb.log
python /root/repo/tools/trace_summ.py /tmp/trb | tail -8
