for shape in "784000 32" "392000 64" "98000 64" "32000 128"; do
for v in "256 8" "512 8" "512 4" "1024 4" "1024 2" "2048 2"; do set -- $v
echo "== shape=$shape cap=$1 rows=$2"
DDL_CR_SMALL_CAP=$1 DDL_CR_SMALL_ROWS=$2 timeout -k 10 60 python3 tools/one_ew.py $shape 2>/dev/null | cut -c1-16 | tr '\n' ' ' || exit 1
echo
done; done
