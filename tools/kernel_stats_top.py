import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) > 1.0:
        print("   %-60s calls %4s avg %9.1f us  tot %8.2f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
