cd /tmp && export TMPDIR=/tmp
for n in 1024 2048; do
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tr$n -o t -- python /root/repo/tools/trace_small.py $n > /tmp/tr$n.log 2>&1 || tail -5 /tmp/tr$n.log
  echo "== n=$n"; python /root/repo/tools/trace_summ.py /tmp/tr$n | grep -v Cijk | tail -12
done
