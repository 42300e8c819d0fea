cd /root/repo
ls tests/test_gpu_*.py | sort -r > /tmp/files_rev.txt
echo "== reversed file order"; timeout 2400 python -X faulthandler -m pytest $(cat /tmp/files_rev.txt | tr '\n' ' ') -m gpu -q -x -p no:cacheprovider 2>&1 | grep -v "Warning\|warnings.warn\|^$" | tail -40
