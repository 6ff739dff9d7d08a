# Round 6, same box: predict_tables (0.4 M genes / 0.6 M rows) with the native output-column passes against the numpy passes of round 5, ms
for r in 1 2 3 4; do for c in 0 1; do GECCO_AMD_TABLES_NUMPY_PASSES=$c python tools/prof_tables.py 2>/dev/null | grep "predict_tables ms" | awk -v c=$c '{printf "numpy_passes=%s %.2f ", c, $3} END{print ""}'; done; done
