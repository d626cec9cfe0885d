import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(r['Name'][:62].ljust(62), r['Calls'].rjust(5), f"{float(r['AverageNs'])/1000:8.2f} us  min {float(r['MinNs'])/1000:7.2f}")
